"""Builds the gfx950 shared library of the step path in-tree:
    python -m red_gym_amd.build
hipcc cross-compiles without a GPU; the resulting libf110_hip.so is git-ignored
but travels with the tree to the GPU box."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'f110_abi.hip')
DEPS = [SRC] + sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'f110_hip.h')]
LIB = os.path.join(HERE, 'libf110_hip.so')

# -ffp-contract=off: the reference's cell / LUT indices and collision decisions are
# products of separately rounded fp64 mul/add; an FMA would change them.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         '-Wno-unused-value']


def build(force=False, verbose=False):
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)):
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + ['-o', LIB, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
