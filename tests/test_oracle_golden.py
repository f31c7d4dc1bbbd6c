"""Pins the CPU oracle (oracle/f110_oracle.c) against vectors generated from the
reference itself (tests/golden/make_golden.py) and the reference's own KATs.

Tolerances: bit-exact (==) wherever the path is pure IEEE arithmetic on given
inputs (ray march, iTTC, constraints, LUT indices, lookup counts, booleans,
indices).  1e-12 where libm (sin/cos/tan/atan2: numpy ships its own kernels) or
BLAS rounding (ndarray.dot in the reference) enters -- see oracle/f110_oracle.c.
"""
import os

import numpy as np
import pytest

import oracle

TOL = 1e-12


@pytest.fixture(scope='module')
def ex_scanner(assets):
    s = oracle.Scanner(1080, 2 * np.pi)
    s.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    return s


def test_map_table_matches_reference(golden, ex_scanner):
    g = golden('g1_scan.npz')
    dt = ex_scanner.map['dt']
    assert dt.shape == (1600, 1600)
    assert np.array_equal(dt[g['ex_dt_rows'], g['ex_dt_cols']], g['ex_dt_vals'])
    assert np.sum(dt) == g['ex_dt_sum']
    assert dt[-1, -1] == g['ex_dt_corner']
    # dt == res*sqrt(integer) exactly
    d2 = np.rint((dt / 0.0625) ** 2)
    assert np.array_equal(dt, 0.0625 * np.sqrt(d2))


def test_scan_example_map_bit_exact(golden, ex_scanner):
    g = golden('g1_scan.npz')
    scans, lk = ex_scanner.scan_batch(g['ex_poses'], return_lookups=True)
    assert np.array_equal(scans, g['ex_scans'])
    assert np.array_equal(lk, g['ex_lookups'])
    for k in range(g['ex_poses'].shape[0]):
        assert np.array_equal(ex_scanner.beam_indices(g['ex_poses'][k]), g['ex_idx'][k])


def test_scan_fov47_and_odd_config(golden, assets):
    g = golden('g1_scan.npz')
    s = oracle.Scanner(1080, 4.7)
    s.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    scans, lk = s.scan_batch(g['ex47_poses'], return_lookups=True)
    assert np.array_equal(scans, g['ex47_scans']) and np.array_equal(lk, g['ex47_lookups'])
    s = oracle.Scanner(271, 4.7, eps=0.001, theta_dis=1500, max_range=12.0)
    s.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    scans, lk = s.scan_batch(g['exx_poses'], return_lookups=True)
    assert np.array_equal(scans, g['exx_scans']) and np.array_equal(lk, g['exx_lookups'])


@pytest.mark.parametrize('name', ['berlin', 'skirk', 'vegas'])
def test_scan_other_maps_and_legacy_fixture(golden, assets, name):
    g = golden('g1_scan.npz')
    s = oracle.Scanner(1080, 4.7)
    s.set_map(os.path.join(assets, 'maps', name + '.yaml'), '.png')
    assert s.map['dt'][-1, -1] == g[name + '_dt_corner']
    scans = s.scan_batch(g[name + '_poses'])
    assert np.array_equal(scans, g[name + '_scans'])
    if name != 'vegas':
        # the reference's own fixture test: MSE < 2 vs the legacy C++ simulator
        # (unittest/scan_sim.py:321-366)
        legacy = golden('legacy_scan.npz')[name]
        mse = np.mean((scans - legacy) ** 2)
        assert mse < 2.


def test_noise_table(golden):
    g = golden('g2_noise.npz')
    for seed in (12345, 0, 7):
        assert np.array_equal(oracle.noise_table(seed, 8), g['seed%d' % seed])


def test_dynamics_kat_from_reference():
    """dynamic_models.py:255-279 test_derivatives ground truth."""
    mu, C_Sf, C_Sr = 1.0489, 21.92 / 1.0489, 21.92 / 1.0489
    lf, lr, h = 0.3048 * 3.793293, 0.3048 * 4.667707, 0.3048 * 2.01355
    m, I = 4.4482216152605 / 0.3048 * 74.91452, 4.4482216152605 * 0.3048 * 1321.416
    p = np.array([mu, C_Sf, C_Sr, lf, lr, h, m, I, -1.066, 1.066, -0.4, 0.4, 7.319, 11.5, -13.6, 50.8, 0, 0])
    f_ks_gt = [16.3475935934250209, 0.4819314886013121, 0.1500000000000000, 5.1464424102339752, 0.2401426578627629]
    f_st_gt = [15.7213512030862397, 0.0925527979719355, 0.1500000000000000, 5.3536773276413925,
               0.0529001056654038, 0.6435589397748606, 0.0313297971641291]
    x_ks = np.array([3.9579422297936526, 0.0391650102771405, 0.0378491427211811, 16.3546957860883566, 0.0294717351052816])
    x_st = np.array([2.0233348142065677, 0.0041907137716636, 0.0197545248559617, 15.7216236334290116,
                     0.0025857914776859, 0.0529001056654038, 0.0033012170610298])
    u = np.array([0.15, 0.63 * 9.81])
    assert np.max(np.abs(oracle.vehicle_dynamics_ks(x_ks, u, p) - f_ks_gt)) < 1e-7  # assertAlmostEqual(.., 0.)
    assert np.max(np.abs(oracle.vehicle_dynamics_st(x_st, u, p) - f_st_gt)) < 1e-7


def test_dynamics_rhs_pid_golden(golden):
    g = golden('g3_dynamics.npz')
    p = oracle.params_vec()
    st, act = g['state'], g['action']
    for i in range(1000):
        assert np.allclose(oracle.vehicle_dynamics_st(st[i], act[i], p), g['f_st'][i], rtol=1e-13, atol=TOL)
        assert np.allclose(oracle.vehicle_dynamics_ks(st[i, :5], act[i], p), g['f_ks'][i], rtol=1e-13, atol=TOL)
        a, sv = oracle.pid(act[i, 1], act[i, 0], st[i, 3], st[i, 2], 3.2, 9.51, 20.0, -5.0)
        assert (a, sv) == (g['pid'][i, 0], g['pid'][i, 1])


@pytest.mark.parametrize('tag,integ', [('rk4', oracle.RK4), ('euler', oracle.EULER)])
def test_update_pose_golden(golden, tag, integ):
    g = golden('g3_dynamics.npz')
    ns, nb, nc = oracle.update_pose_batch(g['state'], g['buf'] * (np.arange(2)[None, :] < g['cnt'][:, None]),
                                          g['cnt'], g['action'], oracle.params_vec(), 0.01, integ)
    assert np.array_equal(nc, g[tag + '_cnt'])
    assert np.array_equal(nb * (np.arange(2)[None, :] < nc[:, None]), g[tag + '_buf'])
    assert np.allclose(ns, g[tag + '_state'], rtol=1e-13, atol=TOL)
    # steer (integrated exactly: no libm) must be bit-exact
    assert np.array_equal(ns[:, 2], g[tag + '_state'][:, 2])


def test_update_pose_sequences_from_reset(golden):
    g = golden('g3_dynamics.npz')
    p = oracle.params_vec()
    n = g['seq_pose'].shape[0]
    st = np.zeros((n, 7))
    st[:, 0:2] = g['seq_pose'][:, 0:2]
    st[:, 4] = g['seq_pose'][:, 2]
    buf, cnt = np.zeros((n, 2)), np.zeros(n, dtype=np.int64)
    for k in range(g['seq_act'].shape[1]):
        st, buf, cnt = oracle.update_pose_batch(st, buf, cnt, g['seq_act'][:, k], p, 0.01, oracle.RK4)
        assert np.allclose(st, g['seq_state'][:, k], rtol=1e-12, atol=1e-11)


def test_gjk_golden(golden):
    g = golden('g4_gjk.npz')
    va = np.array([oracle.get_vertices(p, 0.58, 0.31) for p in g['pose_a']])
    vb = np.array([oracle.get_vertices(p, 0.58, 0.31) for p in g['pose_b']])
    assert np.allclose(va, g['verts_a'], rtol=0, atol=TOL) and np.allclose(vb, g['verts_b'], rtol=0, atol=TOL)
    # booleans on the reference's own vertices: exact
    hit = np.array([oracle.collision(g['verts_a'][i], g['verts_b'][i]) for i in range(len(g['hit']))])
    assert np.array_equal(hit, g['hit'])
    assert 0.2 < hit.mean() < 0.8
    for A in (2, 3, 4):
        poses = g['multi%d_poses' % A]
        for k in range(poses.shape[0]):
            allv = np.stack([oracle.get_vertices(poses[k, a], 0.58, 0.31) for a in range(A)])
            col, idx = oracle.collision_multiple(allv)
            assert np.array_equal(col, g['multi%d_col' % A][k]) and np.array_equal(idx, g['multi%d_idx' % A][k])


def test_gjk_kat_from_reference(golden):
    """collision_models.py:306-324 (np.random.seed(1234) stream) answers."""
    g = golden('g4_gjk.npz')
    np.random.seed(1234)
    v1 = np.asarray([[4, 11.], [5, 5], [9, 9], [10, 10]])
    kat = [v1 + np.random.normal(size=v1.shape) / 100. for _ in range(6)] + [v1 + 10.]
    assert np.array_equal(np.stack(kat), g['kat_verts'])
    col, idx = oracle.collision_multiple(np.stack(kat))
    assert np.array_equal(col, [1., 1., 1., 1., 1., 1., 0.])
    assert np.array_equal(idx, [5., 5., 5., 5., 5., 4., -1.])
    for _ in range(1000):  # test_random_collision
        a = v1 + np.random.normal(size=v1.shape) / 100.
        b = v1 + np.random.normal(size=v1.shape) / 100.
        assert oracle.collision(a, b)


def test_ttc_golden(golden):
    g = golden('g5_ttc.npz')
    s = oracle.Scanner(1080, 2 * np.pi)
    assert np.allclose(s.scan_angles, g['scan_angles'], rtol=0, atol=1e-15)
    assert np.allclose(s.beam_cosines, g['cosines'], rtol=0, atol=1e-15)
    assert np.allclose(s.side_distances, g['side_distances'], rtol=1e-14, atol=0)
    # use the reference's tables so that the comparison itself is exact
    s.beam_cosines[:] = g['cosines']
    s.side_distances[:] = g['side_distances']
    scans = g['scans_f32'].astype(np.float64)
    for i in range(scans.shape[0]):
        if g['ov_beam'][i] >= 0:
            scans[i, g['ov_beam'][i]] = g['ov_val'][i]
    hit = np.array([s.check_ttc(scans[i], g['vel'][i]) for i in range(scans.shape[0])])
    assert np.array_equal(hit, g['hit'])
    assert 0.1 < hit.mean() < 0.9


def test_raycast_golden(golden):
    g = golden('g6_raycast.npz')
    s = oracle.Scanner(1080, 2 * np.pi)
    s.scan_angles[:] = g['scan_angles']
    scans_in = g['scans_in_f32'].astype(np.float64)
    n = scans_in.shape[0]
    for i in range(n):
        lo, hi = s.blocked_view_indices(g['ego'][i], g['verts'][i])
        assert (lo, hi) == tuple(g['span'][i])
        out = s.ray_cast(g['ego'][i], scans_in[i], g['verts'][i])
        assert np.allclose(out, g['scans_out'][i], rtol=0, atol=1e-11)
        assert np.array_equal(out != scans_in[i], g['scans_out'][i] != scans_in[i])
    assert (g['scans_out'] != scans_in).any(axis=1).mean() > 0.9


def _mk_env(assets, A, noise_steps):
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    return oracle.Env(sc, A, noise=oracle.noise_table(12345, noise_steps))


def test_sim_step_one_agent_golden(golden, assets):
    g = golden('g7_sim.npz')
    T = g['a1_actions'].shape[0]
    env = _mk_env(assets, 1, T + 2)
    env.reset(g['a1_start'])  # Simulator.reset + one zero-action step
    scan_at = dict(zip(g['a1_scan_steps'].tolist(), g['a1_scans']))
    for k in range(T):
        o = env.sim_step(g['a1_actions'][k])
        assert np.allclose(o['state'], g['a1_states'][k], rtol=0, atol=1e-9), k
        assert np.array_equal(o['collisions'], g['a1_collisions'][k]), k
        if k in scan_at:
            assert np.allclose(o['scans'][0], scan_at[k], rtol=0, atol=1e-9), k
    assert g['a1_collisions'].any()


def test_sim_step_two_agents_golden(golden, assets):
    g = golden('g7_sim.npz')
    T = g['a2_actions'].shape[0]
    env = _mk_env(assets, 2, T + 2)
    env.reset(g['a2_start'])
    scan_at = dict(zip(g['a2_scan_steps'].tolist(), g['a2_scans']))
    for k in range(T):
        o = env.sim_step(g['a2_actions'][k])
        assert np.allclose(o['state'], g['a2_states'][k], rtol=0, atol=1e-9), k
        assert np.array_equal(o['collisions'], g['a2_collisions'][k]), k
        assert np.array_equal(o['collision_idx'], g['a2_collision_idx'][k]), k
        if k in scan_at:
            assert np.allclose(o['scans'], scan_at[k], rtol=0, atol=1e-9), k
    assert g['a2_collisions'].any() and (g['a2_collision_idx'] >= 0).any()


def test_env_closed_loop_golden(golden, assets):
    """F110Env 2-lap run driven by the recorded pure-pursuit actions
    (examples/waypoint_follow.py): 3329 steps, 2 laps, no collision."""
    g = golden('g8_env.npz')
    T = g['actions'].shape[0]
    assert T == 3329
    env = _mk_env(assets, 1, T + 2)
    o = env.reset(g['start'])
    ro = g['reset_obs']
    assert np.allclose([o['state'][0, 0], o['state'][0, 1], o['state'][0, 4], o['state'][0, 3]], ro[:4], atol=1e-12)
    assert o['lap_times'][0] == ro[4] and o['lap_counts'][0] == ro[5]
    assert np.allclose(o['scans'][0], g['scans'][0], rtol=0, atol=1e-9)
    scan_at = dict(zip(g['scan_steps'].tolist()[1:], g['scans'][1:]))
    for k in range(T):
        o = env.step(g['actions'][k][None, :])
        assert np.allclose(o['state'][0], g['state'][k], rtol=0, atol=1e-7), k
        assert o['collisions'][0] == g['col'][k]
        assert o['toggles'][0] == g['toggle'][k], k
        assert o['lap_counts'][0] == g['lap_c'][k]
        assert o['lap_times'][0] == g['lap_t'][k]
        assert o['done'] == bool(g['done'][k]), k
        if k in scan_at:
            assert np.allclose(o['scans'][0], scan_at[k], rtol=0, atol=1e-7), k
    assert o['done'] and o['lap_counts'][0] == 2 and o['current_time'] == g['final_time']


def test_env_closed_loop_two_agents_golden(golden, assets):
    """The oracle's _check_done with A > 1 pinned by the reference's own 2-agent run to all(toggles >= 4) (g9):
    the ego's start rotation on every car (f110_env.py:219-221), frozen lap times, done only through all()."""
    g = golden('g9_env2.npz')
    T = g['actions'].shape[0]
    env = _mk_env(assets, 2, T + 2)
    o = env.reset(g['start'])
    assert np.allclose(o['state'][:, [0, 1, 4]], g['reset_obs'], atol=1e-12)
    scan_at = dict(zip(g['scan_steps'].tolist(), g['scans']))
    for k in range(T):
        o = env.step(g['actions'][k])
        assert np.allclose(o['state'][:, 0], g['x'][k], rtol=0, atol=1e-7) and np.allclose(o['state'][:, 1], g['y'][k], rtol=0, atol=1e-7), k
        assert np.array_equal(o['collisions'], g['col'][k]), k
        assert np.array_equal(o['toggles'], g['toggle'][k]), k
        assert np.array_equal(o['lap_counts'], g['lap_c'][k]) and np.array_equal(o['lap_times'], g['lap_t'][k]), k
        assert o['done'] == bool(g['done'][k]), k
        if k in scan_at:
            assert np.allclose(o['scans'], scan_at[k], rtol=0, atol=1e-7), k
    assert o['done'] and (g['toggle'][-1] >= 4).all() and g['toggle'][-1].max() >= 5
