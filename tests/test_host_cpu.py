"""Host-side logic that needs no GPU: integrator mapping and its error, map path
resolution, the noise table, workload generation, the planner restatement, the
third-party stand-ins, and that the product refuses to run without a HIP device."""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import yaml


def test_integrator_enum_and_error():
    from red_gym_amd.base_classes import Integrator, integrator_code
    assert integrator_code(Integrator.RK4) == 1 and integrator_code(Integrator.Euler) == 2
    assert integrator_code(1) == 1 and integrator_code(2) == 2

    class Other(object):  # the reference's own enum has the same member names
        name = 'Euler'
    assert integrator_code(Other()) == 2
    with pytest.raises(SyntaxError, match='Invalid Integrator'):
        integrator_code(3)


def test_map_path_resolution(assets):
    from red_gym_amd.vec_env import resolve_map_path
    # f110_env.py:106-118
    assert resolve_map_path('berlin') == os.path.join(assets, 'maps', 'berlin.yaml')
    assert resolve_map_path('/x/y/custom') == '/x/y/custom.yaml'


def test_noise_table_is_numpy_stream(golden):
    from red_gym_amd.engine import NoiseTable
    g = golden('g2_noise.npz')
    nt = NoiseTable(12345, 1080)
    assert np.array_equal(nt.ensure(3), g['seed12345'][:3])
    rows = nt.ensure(8)  # growing continues the same generator
    assert np.array_equal(rows, g['seed12345'])
    assert nt.ensure(4) is rows
    # laser_models.py:554-580 test_rng: same seed, same scan noise
    assert np.array_equal(NoiseTable(12345, 1080).ensure(2), rows[:2]) and not np.array_equal(rows[0], rows[1])


def test_ziggurat_tables_and_the_generator_model(golden):
    """The device noise generator (csrc/f110_noise.h) is compiled with csrc/f110_ziggurat.h.  Its CPU model --
    tools/make_ziggurat_tables.py: PCG64 + ziggurat restated with those committed tables -- reproduces the golden rows the
    reference's rng use produced (g2) and NumPy's stream for other seeds, wedge and tail branches included."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import make_ziggurat_tables as z
    tables = z.read_header()
    assert len(tables[0]) == len(tables[1]) == len(tables[2]) == 256 and tables[2][0] == 1.0
    g = golden('g2_noise.npz')['seed12345']
    assert np.array_equal(z.emulate_rows(12345, 3, 1080, 0.01, tables), g[:3])
    stats = {}
    for seed in (0, 7, 2 ** 40 + 3):
        rng = np.random.default_rng(seed)
        ref = np.stack([rng.normal(0., 0.01, size=1080) for _ in range(2)])
        assert np.array_equal(z.emulate_rows(seed, 2, 1080, 0.01, tables, stats), ref), seed
    assert stats.get('wedge', 0) > 10


def test_beam_tables_match_reference(golden):
    from red_gym_amd.engine import DEFAULT_PARAMS, beam_tables
    g = golden('g5_ttc.npz')
    ang, cosv, side = beam_tables(1080, 2 * np.pi, DEFAULT_PARAMS)
    assert np.array_equal(ang, g['scan_angles']) and np.array_equal(cosv, g['cosines'])
    assert np.array_equal(side, g['side_distances'])


def test_workload_is_deterministic_and_sharded():
    from red_gym_amd import workload
    p0 = workload.spawn_poses(128, 2, rank=0)
    assert np.array_equal(p0, workload.spawn_poses(128, 2, rank=0))
    p1 = workload.spawn_poses(128, 2, rank=1)
    assert p0.shape == (128, 2, 3) and not np.array_equal(p0, p1)
    # second car ~1.5 m behind the first along the raceline (before jitter sigma 0.2)
    d = np.hypot(p0[:, 0, 0] - p0[:, 1, 0], p0[:, 0, 1] - p0[:, 1, 1])
    assert 0.6 < np.median(d) < 2.4
    a = workload.action_pool(4, 128, 2, rank=3)
    assert a.shape == (4, 128, 2, 2) and np.abs(a[..., 0]).max() <= 0.4189 and 0 <= a[..., 1].min() and a[..., 1].max() <= 8


def test_planner_reproduces_reference_actions(golden, assets):
    """examples/waypoint_follow.py planner: the action recorded at step k was planned
    from the pose observed after step k-1."""
    from oracle.planner import PurePursuitPlanner
    conf = Namespace(**yaml.safe_load(open(os.path.join(assets, 'config_example_map.yaml'))))
    conf.wpt_path = os.path.join(assets, 'example_waypoints.csv')
    pl = PurePursuitPlanner(conf, 0.17145 + 0.15875)
    g = golden('g8_env.npz')
    ro = g['reset_obs']
    sp, st = pl.plan(ro[0], ro[1], ro[2], 0.82461887897713965, 1.375)
    assert abs(st - g['actions'][0, 0]) < 1e-12 and abs(sp - g['actions'][0, 1]) < 1e-12
    for k in range(1, 3329, 7):
        sp, st = pl.plan(g['x'][k - 1], g['y'][k - 1], g['theta'][k - 1], 0.82461887897713965, 1.375)
        assert abs(st - g['actions'][k, 0]) < 1e-12 and abs(sp - g['actions'][k, 1]) < 1e-12


def test_compat_standins_and_gym_registration():
    from red_gym_amd import compat
    compat.install_missing()
    import gym
    import f110_gym  # noqa: F401
    from f110_gym.envs.base_classes import Integrator
    assert Integrator.RK4.value == 1
    from numba import njit

    @njit(cache=True)
    def f(x):
        return x + 1
    assert f(1) == 2
    if not hasattr(gym, '__version__'):  # the stand-in registry
        with pytest.raises(KeyError):
            gym.make('f110_gym:nope-v0')


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a HIP device is present')
    from red_gym_amd.engine import Engine
    with pytest.raises(RuntimeError, match='no CPU path'):
        Engine(num_envs=1, num_agents=1)
    from red_gym_amd import F110Env
    with pytest.raises(RuntimeError):
        F110Env(map='berlin', num_agents=1)


def test_usable_cores():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= 64


def test_map_name_resolution_matches_reference(assets):
    """f110_env.py:106-118: berlin / skirk / levine are packaged; an explicit 'vegas' is ./vegas.yaml; only an
    absent `map` keyword selects the packaged vegas.  'levine' ships no image (there as here): clear error."""
    from red_gym_amd.vec_env import resolve_map_path
    from red_gym_amd.maps import load_map
    for name in ('berlin', 'skirk', 'levine'):
        assert resolve_map_path(name) == os.path.join(assets, 'maps', name + '.yaml')
    assert resolve_map_path('vegas') == 'vegas.yaml'
    assert resolve_map_path('/tmp/some/track') == '/tmp/some/track.yaml'
    assert resolve_map_path(None) == os.path.join(assets, 'maps', 'vegas.yaml')
    with pytest.raises(FileNotFoundError, match='levine.png'):
        load_map(resolve_map_path('levine'), '.png')


def test_log1p_model_is_the_c_librarys_log1p():
    """tools/log1p_model.py restates glibc's log1p (the text of log1p_glibc in csrc/f110_noise.h, the ziggurat's tail branch)
    in Python; it must be math.log1p bit for bit on the machine's libm -- uniform arguments, the edges of every branch."""
    import math
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import log1p_model as m
    rng = np.random.default_rng(3)
    us = np.concatenate([rng.random(40000), 1.0 - 2.0 ** -rng.integers(1, 54, 500), 2.0 ** -rng.uniform(1, 60, 3000),
                         [0.0, 2.0 ** -53, 1.0 - 2.0 ** -53, 0.5, 0.2929, 0.29289321881345254, 0.2928932188134525]])
    for u in us:
        assert m.log1p_glibc(-float(u)) == math.log1p(-float(u)), u
