"""No-GPU checks of the shipped library: it builds, loads, exports every symbol
that include/f110_hip.h declares, and its host-only entry point (exact EDT)
reproduces scipy's distance transform on every bundled map."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from red_gym_amd import _lib, build
from red_gym_amd.maps import load_map

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    build.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'f110_hip.h')).read()
    declared = set(re.findall(r'\b(f110_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    hdr = open(os.path.join(ROOT, 'include', 'f110_hip.h')).read()
    body = hdr[hdr.index('typedef struct {', hdr.index('Caller-owned device buffers')):hdr.index('} f110_buffers;')]
    fields = re.findall(r'\*\s*([a-z_0-9]+);', body)
    assert fields == _lib.BUFFER_FIELDS
    assert C.sizeof(_lib.Config) == 8 * 4 + 5 * 8 + 18 * 8
    # the limits the Python mirror checks against are the header's
    for name in ('F110_MAX_AGENTS', 'F110_MAX_NOISE_SLOTS', 'F110_MAX_MAPS', 'F110_NUM_PARAMS'):
        m = re.search(r'#define\s+%s\s+\(?(\d+)' % name, hdr)
        assert m and int(m.group(1)) == getattr(_lib, name), name


def test_error_reporting_without_gpu(lib):
    # argument validation happens before any HIP call
    rc = lib.f110_create(None, None)
    assert rc == _lib.E_INVALID and b'null' in lib.f110_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    out = np.zeros((4, 4), dtype=np.uint32)
    ones = np.ones((4, 4), dtype=np.uint8)
    rc = lib.f110_edt_squared(ones.ctypes.data_as(C.c_void_p), 4, 4, out.ctypes.data_as(C.c_void_p))
    assert rc == _lib.E_INVALID  # no occupied cell


@pytest.mark.parametrize('rel', ['example_map.yaml', 'maps/berlin.yaml', 'maps/skirk.yaml', 'maps/vegas.yaml'])
def test_exact_edt_equals_scipy(lib, assets, rel):
    from scipy.ndimage import distance_transform_edt as edt
    from red_gym_amd.engine import edt_squared
    m = load_map(os.path.join(assets, rel), '.png')
    d2 = edt_squared(m.free)
    ref = m.resolution * edt(m.free.astype(np.float64) * 255.)  # laser_models.py:52,425
    assert np.array_equal(m.resolution * np.sqrt(d2.astype(np.float64)), ref)


def test_exact_edt_small_cases(lib):
    from scipy.ndimage import distance_transform_edt as edt
    from red_gym_amd.engine import edt_squared
    rng = np.random.default_rng(5)
    for shape in [(1, 1), (1, 7), (9, 1), (5, 8), (33, 17), (64, 64)]:
        for dens in (0.02, 0.3, 0.9):
            free = (rng.uniform(size=shape) > dens).astype(np.uint8)
            free.flat[rng.integers(free.size)] = 0
            d2 = edt_squared(free)
            assert np.array_equal(np.sqrt(d2.astype(np.float64)), edt(free))


def test_map_loader_matches_oracle_loader(assets):
    import oracle
    for rel in ['example_map.yaml', 'maps/skirk.yaml']:
        m = load_map(os.path.join(assets, rel), '.png')
        o = oracle.load_map(os.path.join(assets, rel), '.png')
        assert (m.height, m.width, m.resolution) == (o['height'], o['width'], o['resolution'])
        assert (m.orig_x, m.orig_y, m.orig_c, m.orig_s) == (o['orig_x'], o['orig_y'], o['orig_c'], o['orig_s'])
        assert np.array_equal(m.free * 255., o['img'])


def test_product_never_imports_oracle():
    """The shipped package, its drop-in shims and the examples must not reference oracle/ (judge rule: no CPU
    fallback, the checker is test infrastructure); bench.py may, in its cpu_baseline leg only."""
    for top in ('red_gym_amd', 'f110_gym', 'weap_util', 'examples', 'include'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(('.py', '.h', '.hip', '.cpp')):
                    src = open(os.path.join(dirpath, f)).read()
                    assert 'import oracle' not in src and 'from oracle' not in src, os.path.join(dirpath, f)
                    assert 'f110_oracle' not in src and 'oracle/_build' not in src, os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    leg = bench[bench.index('def cpu_baseline('):bench.index('class Ranks(')]
    rest = bench.replace(leg, '')
    assert 'import oracle' in leg and 'import oracle' not in rest and 'from oracle' not in rest


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No fallback: without libf110_hip.so the import path raises and names the build command."""
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope' / 'libf110_hip.so'))
    with pytest.raises(ImportError, match='no CPU fallback'):
        _lib.load()
