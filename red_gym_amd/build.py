"""Builds the gfx950 shared library of the step path in-tree:
    python -m red_gym_amd.build [--force]
hipcc cross-compiles without a GPU; the resulting libf110_hip.so is git-ignored
but travels with the tree to the GPU box.  The library is five translation units
(csrc/f110_internal.h lists them): they are compiled in parallel, each only when it or
a header it includes has changed, and linked into the one .so."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
UNITS = ['f110_handle', 'f110_maps', 'f110_noise_abi', 'f110_step', 'f110_consumers']
HEADERS = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'f110_hip.h')]
LIB = os.path.join(HERE, 'libf110_hip.so')
OBJ_DIR = os.path.join(HERE, 'build')

# -ffp-contract=off: the reference's cell / LUT indices and collision decisions are
# products of separately rounded fp64 mul/add; an FMA would change them.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-value']


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=LIB, obj_dir=OBJ_DIR):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(obj_dir, exist_ok=True)
    jobs = []
    for u in UNITS:
        src, obj = os.path.join(CSRC, u + '.hip'), os.path.join(obj_dir, u + '.o')
        if force or not _newer(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ['-c', '-o', obj, src])
    if not jobs and not force and _newer(lib, [os.path.join(obj_dir, u + '.o') for u in UNITS]):
        return lib

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + [os.path.join(obj_dir, u + '.o') for u in UNITS])
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
