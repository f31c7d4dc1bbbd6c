"""Step-path parity on the GPU: f110_reset / f110_step through the C ABI against
the oracle's Env and the reference-generated trajectories (g7, g8)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def _mk_oracle_env(assets, A, noise_steps, integrator=oracle.RK4):
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    return oracle.Env(sc, A, noise=oracle.noise_table(12345, noise_steps), integrator=integrator)


def _mk_scanner(assets):
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    return sc


def _vec(assets, B, A, **kw):
    from red_gym_amd import F110VecEnv
    kw.setdefault('autoreset', False)
    return F110VecEnv(B, map=os.path.join(assets, 'example_map'), map_ext='.png', num_agents=A,
                      keep_f64_scans=True, **kw)


def test_sim_step_one_agent_golden(golden, assets):
    g = golden('g7_sim.npz')
    env = _vec(assets, 1, 1)
    env.reset(g['a1_start'][None])
    T = g['a1_actions'].shape[0]
    scan_at = dict(zip(g['a1_scan_steps'].tolist(), g['a1_scans']))
    for k in range(T):
        obs, _, done, info = env.step(g['a1_actions'][k][None])
        assert np.allclose(_np(env.state)[0], g['a1_states'][k], rtol=0, atol=1e-9), k
        assert np.array_equal(_np(obs['collisions'])[0].astype(np.float64), g['a1_collisions'][k]), k
        if k in scan_at:
            assert np.allclose(_np(obs['scans_f64'])[0, 0], scan_at[k], rtol=0, atol=1e-9), k
            assert np.allclose(_np(obs['scans'])[0, 0], scan_at[k], rtol=0, atol=1e-5), k
    env.close()


def test_sim_step_two_agents_golden(golden, assets):
    g = golden('g7_sim.npz')
    env = _vec(assets, 1, 2)
    env.reset(g['a2_start'][None])
    T = g['a2_actions'].shape[0]
    scan_at = dict(zip(g['a2_scan_steps'].tolist(), g['a2_scans']))
    for k in range(T):
        obs, _, done, info = env.step(g['a2_actions'][k][None])
        assert np.allclose(_np(env.state)[0], g['a2_states'][k], rtol=0, atol=1e-9), k
        assert np.array_equal(_np(obs['collisions'])[0].astype(np.float64), g['a2_collisions'][k]), k
        assert np.array_equal(_np(info['collision_idx'])[0].astype(np.float64), g['a2_collision_idx'][k]), k
        if k in scan_at:
            assert np.allclose(_np(obs['scans_f64'])[0], scan_at[k], rtol=0, atol=1e-9), k
    env.close()


def test_env_closed_loop_golden(golden, assets):
    """Config 1: the 2-lap waypoint-follow run (3329 steps, 2 laps, no collision)
    replayed through the single-env F110Env facade."""
    from red_gym_amd import F110Env, Integrator
    g = golden('g8_env.npz')
    env = F110Env(map=os.path.join(assets, 'example_map'), map_ext='.png', num_agents=1, timestep=0.01,
                  integrator=Integrator.RK4)
    obs, r, done, info = env.reset(g['start'])
    ro = g['reset_obs']
    assert np.allclose([obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0]], ro[:4], atol=1e-12)
    assert obs['lap_times'][0] == ro[4] and obs['lap_counts'][0] == ro[5] and r == 0.01 and not done
    assert isinstance(obs['poses_x'], list) and isinstance(obs['poses_x'][0], float)
    assert obs['scans'][0].dtype == np.float64 and obs['scans'][0].shape == (1080,)
    assert np.allclose(obs['scans'][0], g['scans'][0], rtol=0, atol=1e-9)
    scan_at = dict(zip(g['scan_steps'].tolist()[1:], g['scans'][1:]))
    T = g['actions'].shape[0]
    laptime = 0.0
    for k in range(T):
        obs, r, done, info = env.step(g['actions'][k][None, :])
        laptime += r
        assert abs(obs['poses_x'][0] - g['x'][k]) < 1e-7 and abs(obs['poses_y'][0] - g['y'][k]) < 1e-7, k
        assert abs(obs['poses_theta'][0] - g['theta'][k]) < 1e-7 and abs(obs['linear_vels_x'][0] - g['vx'][k]) < 1e-7, k
        assert obs['collisions'][0] == g['col'][k], k
        assert env.toggle_list[0] == g['toggle'][k], k
        assert obs['lap_counts'][0] == g['lap_c'][k] and obs['lap_times'][0] == g['lap_t'][k], k
        assert done == bool(g['done'][k]), k
        assert bool(info['checkpoint_done'][0]) == (g['toggle'][k] >= 4)
        if k in scan_at:
            assert np.allclose(obs['scans'][0], scan_at[k], rtol=0, atol=1e-7), k
    assert done and obs['lap_counts'][0] == 2 and abs(laptime - 33.29) < 1e-9
    env.close()


@pytest.mark.parametrize('A,integ', [(1, 'RK4'), (2, 'RK4'), (3, 'Euler')])
def test_batched_random_rollout_vs_oracle(assets, A, integ):
    """B envs with random actions for 60 steps, every env compared with an independent
    oracle Env: collisions / collision_idx / toggles / lap counts exact, state and
    scans to 1e-9."""
    from red_gym_amd import Integrator
    B, T = 24, 60
    rl = np.loadtxt(os.path.join(assets, 'example_waypoints.csv'), delimiter=';', skiprows=3)
    rng = np.random.default_rng(100 + A)
    poses = np.zeros((B, A, 3))
    for b in range(B):
        k = (97 * b) % rl.shape[0]
        for a in range(A):
            kk = (k - 8 * a) % rl.shape[0]  # followers ~1.5 m behind along the raceline
            poses[b, a] = [rl[kk, 1] + rng.normal(0, 0.1), rl[kk, 2] + rng.normal(0, 0.1),
                           rl[kk, 3] + np.pi / 2 + rng.normal(0, 0.1)]
    poses[0, :, :2] = poses[0, 0, :2]  # identical positions: GJK hit from the first step
    env = _vec(assets, B, A, integrator=getattr(Integrator, integ))
    ors = [_mk_oracle_env(assets, A, T + 2, oracle.RK4 if integ == 'RK4' else oracle.EULER) for _ in range(B)]
    env.reset(poses)
    oo = [ors[b].reset(poses[b]) for b in range(B)]

    def compare(obs, info, done, oo, k):
        st = _np(env.state)
        for b in range(B):
            assert np.allclose(st[b], oo[b]['state'], rtol=0, atol=1e-9), (k, b)
            assert np.array_equal(_np(obs['collisions'])[b].astype(np.float64), oo[b]['collisions']), (k, b)
            assert np.array_equal(_np(info['collision_idx'])[b].astype(np.float64), oo[b]['collision_idx']), (k, b)
            assert np.array_equal(_np(info['toggles'])[b].astype(np.float64), oo[b]['toggles']), (k, b)
            assert np.array_equal(_np(obs['lap_counts'])[b].astype(np.float64), oo[b]['lap_counts']), (k, b)
            assert np.array_equal(_np(obs['lap_times'])[b], oo[b]['lap_times']), (k, b)
            assert bool(_np(done)[b]) == oo[b]['done'], (k, b)
            assert np.allclose(_np(obs['scans_f64'])[b], oo[b]['scans'], rtol=0, atol=1e-9), (k, b)
    obs, _, done, info = env._result()
    compare(obs, info, done, oo, -1)
    hits = 0
    for k in range(T):
        act = np.stack([rng.uniform(-0.4189, 0.4189, (B, A)), rng.uniform(0, 8, (B, A))], axis=2)
        obs, _, done, info = env.step(act)
        oo = [ors[b].step(act[b]) for b in range(B)]
        compare(obs, info, done, oo, k)
        hits += int(_np(obs['collisions']).sum())
    assert hits > 0 or A == 1  # identical start positions guarantee a GJK hit for A > 1
    env.close()


def test_autoreset_and_masked_reset(assets):
    """done envs restart from their spawn pose on the next step (reset + zero-action
    step, f110_env.py:304-347); a masked reset leaves the other envs untouched."""
    B = 8
    rl = np.loadtxt(os.path.join(assets, 'example_waypoints.csv'), delimiter=';', skiprows=3)
    poses = np.stack([[rl[(97 * b) % 783, 1], rl[(97 * b) % 783, 2], rl[(97 * b) % 783, 3] + np.pi / 2] for b in range(B)])[:, None, :]
    env = _vec(assets, B, 1, autoreset=True)
    ors = [_mk_oracle_env(assets, 1, 400) for _ in range(B)]
    env.reset(poses)
    oo = [ors[b].reset(poses[b]) for b in range(B)]
    pending = np.zeros(B, dtype=bool)
    resets = 0
    for k in range(150):
        act = np.zeros((B, 1, 2))
        act[:, 0, 0] = 0.3   # drive into the wall
        act[:, 0, 1] = 6.0
        obs, _, done, info = env.step(act)
        for b in range(B):
            if pending[b]:
                oo[b] = ors[b].reset(poses[b])
                resets += 1
            else:
                oo[b] = ors[b].step(act[b])
            pending[b] = oo[b]['done']
            assert bool(_np(done)[b]) == oo[b]['done'], (k, b)
            assert np.allclose(_np(env.state)[b], oo[b]['state'], rtol=0, atol=1e-9), (k, b)
            assert np.allclose(_np(obs['scans_f64'])[b], oo[b]['scans'], rtol=0, atol=1e-9), (k, b)
            assert _np(info['current_time'])[b] == oo[b]['current_time']
    assert resets >= B
    # masked reset: env 2 only
    import torch
    before = _np(env.state).copy()
    mask = torch.zeros(B, dtype=torch.uint8)
    mask[2] = 1
    env.reset(poses, mask=mask)
    after = _np(env.state)
    o2 = ors[2].reset(poses[2])
    assert np.allclose(after[2], o2['state'], atol=1e-9)
    keep = [b for b in range(B) if b != 2]
    assert np.array_equal(after[keep], before[keep])
    env.close()


def test_reset_errors(assets):
    env = _vec(assets, 2, 2)
    with pytest.raises(ValueError, match='Number of poses'):
        env.reset(np.zeros((2, 3, 3)))
    with pytest.raises(IndexError):
        env.update_params(env.params, index=5)
    env.close()


def test_per_agent_update_params(assets):
    """Simulator.update_params(params, agent_idx) (base_classes.py:507-527): only that
    agent's dynamics change; the oracle models it with a second env using the new params."""
    from red_gym_amd import Integrator
    B, A = 4, 2
    rl = np.loadtxt(os.path.join(assets, 'example_waypoints.csv'), delimiter=';', skiprows=3)
    poses = np.stack([[[rl[(97 * b) % 783, 1], rl[(97 * b) % 783, 2], rl[(97 * b) % 783, 3] + np.pi / 2],
                       [rl[(97 * b + 300) % 783, 1], rl[(97 * b + 300) % 783, 2], rl[(97 * b + 300) % 783, 3] + np.pi / 2]]
                      for b in range(B)])
    heavy = dict(oracle.DEFAULT_PARAMS, m=5.0, I=0.08, mu=0.7)
    env = _vec(assets, B, A, integrator=Integrator.RK4)
    env.update_params(heavy, index=1)
    with pytest.raises(IndexError):
        env.update_params(heavy, index=2)
    env.reset(poses)
    # far-apart cars do not interact, so each agent can be checked against a 1-agent oracle env
    o0 = [_mk_oracle_env(assets, 1, 64) for _ in range(B)]
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    o1 = [oracle.Env(sc, 1, params=heavy, noise=oracle.noise_table(12345, 64)) for _ in range(B)]
    for b in range(B):
        o0[b].reset(poses[b, 0:1]); o1[b].reset(poses[b, 1:2])
    rng = np.random.default_rng(5)
    for k in range(40):
        act = np.stack([rng.uniform(-0.3, 0.3, (B, A)), rng.uniform(2, 7, (B, A))], axis=2)
        env.step(act)
        st = _np(env.state)
        for b in range(B):
            assert np.allclose(st[b, 0], o0[b].step(act[b, 0:1])['state'][0], rtol=0, atol=1e-9), (k, b)
            assert np.allclose(st[b, 1], o1[b].step(act[b, 1:2])['state'][0], rtol=0, atol=1e-9), (k, b)
    assert not np.allclose(st[:, 0, 3], st[:, 1, 3])
    env.close()


@pytest.mark.parametrize('source', ['numpy', 'device'])
def test_noise_table_grows_before_it_is_exhausted(assets, source):
    """A car that keeps driving past the rows the noise table holds: more are there before any car needs them
    (engine._ensure_noise) -- drawn by NumPy on the host and uploaded, or produced on the device."""
    from red_gym_amd.engine import Engine
    import torch
    e = Engine(num_envs=2, num_agents=1, noise_steps=4, keep_f64_scans=True, noise_source=source)
    e.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    poses = np.array([[[0.7, 0.0, 1.37079632679]], [[0.7, 0.0, 1.37079632679]]])
    e.reset(torch.as_tensor(poses))
    orc = _mk_oracle_env(assets, 1, 64)
    oo = orc.reset(poses[0])
    act = np.zeros((2, 1, 2))
    act[:, 0, 1] = 1.0
    for k in range(30):
        e.step(torch.as_tensor(act))
        oo = orc.step(act[0])
        assert np.allclose(_np(e.t['scans_f64'])[0, 0], oo['scans'][0], rtol=0, atol=1e-9), k
    assert e._noise_rows >= 32 and int(e.t['noise_step'].max()) == 31 and e.device_errors() == 0
    e.close()


def test_update_map_switches_tables(assets, golden):
    env = _vec(assets, 1, 1, noise_std=0)
    g = golden('g1_scan.npz')
    assert np.array_equal(_np(env.eng.scan(g['ex_poses'][:3])), g['ex_scans'][:3])
    env.update_map(os.path.join(assets, 'maps', 'berlin.yaml'), '.png')
    e47 = _np(env.eng.scan(g['berlin_poses'][:2]))  # fov 2pi here, so only compare with the oracle
    s = oracle.Scanner(1080, 2 * np.pi)
    s.set_map(os.path.join(assets, 'maps', 'berlin.yaml'), '.png')
    assert np.array_equal(e47, s.scan_batch(g['berlin_poses'][:2]))
    env.close()


def test_hipgraph_replay_equals_eager(assets):
    """f110_step is capture-safe (no allocation, no synchronisation): a HIP graph of
    planner + step replays to the same bits as eager launches."""
    import torch
    from red_gym_amd import workload
    B = 256
    rl = np.loadtxt(os.path.join(assets, 'example_waypoints.csv'), delimiter=';', skiprows=3)
    wp = torch.as_tensor(np.ascontiguousarray(rl[:, [1, 2, 5]]), device='cuda')
    poses = workload.spawn_poses(B, 1)
    tlad, vgain = 0.82461887897713965, 1.375
    e1, e2 = _vec(assets, B, 1, autoreset=True), _vec(assets, B, 1, autoreset=True)
    e1.reset(poses); e2.reset(poses)
    e2.capture_step(policy=lambda env, out: env.eng.pure_pursuit(wp, tlad, vgain, out=out))
    for k in range(50):
        e1.step(e1.pure_pursuit(wp, tlad, vgain))
        e2.step_graph()
    torch.cuda.synchronize()
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.eng.t['scans'], e2.eng.t['scans'])
    assert torch.equal(e1.eng.t['noise_step'], e2.eng.t['noise_step']) and int(e1.eng.t['noise_step'].max()) > 1
    assert float(e2.state[:, 0, 3].mean()) > 3.0  # the fleet is racing
    e1.close(); e2.close()


@pytest.mark.parametrize('how', ['nodes', 'capture'])
@pytest.mark.parametrize('A', [1, 2])
def test_library_graph_replay_equals_eager(assets, how, A):
    """f110_graph_create builds the step as a HIP graph itself (explicit kernel nodes / a private stream capture):
    replays give the bits of eager steps, the node count is the step's launch count, a change of the handle's launch
    epoch makes f110_graph_launch refuse the stale graph (F110_E_INVALID) and step_lib_graph re-build it."""
    import ctypes as C
    import torch
    from red_gym_amd import _lib, workload
    B = 128
    poses = workload.spawn_poses(B, A)
    acts = torch.as_tensor(workload.action_pool(8, B, A), device='cuda')
    e1, e2 = _vec(assets, B, A, autoreset=True), _vec(assets, B, A, autoreset=True)
    e1.reset(poses); e2.reset(poses)
    buf = e2.build_step_graph(how)
    n_nodes = e2.lib_graph_info()
    assert 2 <= n_nodes <= 5
    keys = [k for k in e1.eng.t if e1.eng.t[k] is not None]
    for k in range(40):
        e1.step(acts[k % 8])
        buf.copy_(acts[k % 8])
        e2.step_lib_graph()
        if k == 20:
            e2.eng.set_scan_stages('*:1')          # bumps the launch epoch
            with pytest.raises(ValueError, match='stale'):
                _lib.check(e2.eng.lib.f110_graph_launch(e2._lg, e2.eng._stream()))
            e1.eng.set_scan_stages('*:1')
    torch.cuda.synchronize()
    for key in keys:
        assert torch.equal(e1.eng.t[key], e2.eng.t[key]), key
    e1.close(); e2.close()


def test_hipgraph_survives_table_changes(assets):
    """A captured step freezes the scan instantiation, the env -> map table and the noise table's base and size.  The
    noise table is re-allocated when a car outlives it (here: a host table of 64 rows that a 71-scan run outgrows), and
    update_map can switch the instantiation (berlin: resolution 0.05, not a power of two): step_graph must notice (launch
    epoch) and re-capture instead of replaying against freed memory.  Scans, state, noise rows `==` eager stepping.
    (The WINDOW of rows a table holds moves without touching the epoch: test_gpu_noise.py, 20 000 replays of one graph.)"""
    import torch
    from red_gym_amd import workload
    B = 64
    poses = workload.spawn_poses(B, 1)
    acts = torch.as_tensor(workload.action_pool(8, B, 1) * np.array([1.0, 0.25]), device='cuda')  # slow: nobody crashes early
    e1, e2 = _vec(assets, B, 1, noise_steps=8, noise_source='numpy'), _vec(assets, B, 1, noise_steps=8, noise_source='numpy')
    e1.reset(poses); e2.reset(poses)
    buf = e2.capture_step()
    ep0 = e2.eng.launch_epoch()
    rows0 = e2.eng._noise_rows
    for k in range(70):                       # 71 scans per car: the host table (64 rows at first) has to double
        e1.step(acts[k % 8])
        buf.copy_(acts[k % 8])
        e2.step_graph()
        assert torch.equal(e1.eng.t['scans_f64'], e2.eng.t['scans_f64']), k
    assert e2.eng._noise_rows > rows0 and e2.eng.launch_epoch() > ep0    # re-allocated: the graph was re-captured
    assert torch.equal(e1.state, e2.state) and int(e2.eng.t['noise_step'].min()) >= 30
    # another map with another scan instantiation, then back
    for y in (os.path.join(assets, 'maps', 'berlin.yaml'), os.path.join(assets, 'example_map.yaml')):
        e1.update_map(y, '.png'); e2.update_map(y, '.png')
        p = np.zeros((B, 1, 3)) if 'berlin' in y else poses
        e1.reset(p); e2.reset(p)
        for k in range(3):
            e1.step(acts[k])
            e2.step_graph(acts[k])
            assert torch.equal(e1.eng.t['scans_f64'], e2.eng.t['scans_f64']), (y, k)
    e1.close(); e2.close()


@pytest.mark.parametrize('cfg', [
    dict(map='maps/berlin', fov=4.7, num_beams=1080, A=2, ego_idx=1, integ='RK4'),      # res 0.05: guarded-reciprocal index path
    dict(map='maps/skirk', fov=2 * np.pi, num_beams=271, A=1, ego_idx=0, integ='Euler'),  # odd beam count (partial chunk)
    dict(map='example_map', fov=3.0, num_beams=64, A=3, ego_idx=2, integ='RK4'),          # one chunk, three agents
    dict(map='maps/vegas', fov=2 * np.pi, num_beams=4096, A=8, ego_idx=7, integ='RK4'),   # both maxima: beams and agents
])
def test_step_path_odd_configs_vs_oracle(assets, cfg):
    """Whole step path (dynamics, scan+noise+iTTC, opponents, GJK, lap logic) on other maps /
    fov / beam counts / ego index, every env against an independent oracle env."""
    from red_gym_amd import F110VecEnv, Integrator
    B, A, T = 6, cfg['A'], 40
    nb = cfg['num_beams']
    map_base = os.path.join(assets, cfg['map'])
    sc = oracle.Scanner(nb, cfg['fov'])
    sc.set_map(map_base + '.yaml', '.png')
    m = sc.map
    rng = np.random.default_rng(77)
    # free-space spawn cells with some clearance
    free = np.argwhere(m['dt'] > 0.6)
    pick = free[rng.choice(len(free), size=B * A, replace=False)]
    poses = np.zeros((B, A, 3))
    poses[..., 0] = (m['orig_x'] + (pick[:, 1] + 0.5) * m['resolution']).reshape(B, A)
    poses[..., 1] = (m['orig_y'] + (pick[:, 0] + 0.5) * m['resolution']).reshape(B, A)
    poses[..., 2] = rng.uniform(0, 6.28, (B, A))
    poses[0, 1:, :2] = poses[0, 0, :2] + 0.2  # overlapping cars: GJK + opponent ray cast at close range
    env = F110VecEnv(B, map=map_base, map_ext='.png', num_agents=A, fov=cfg['fov'], num_beams=nb,
                     ego_idx=cfg['ego_idx'], integrator=getattr(Integrator, cfg['integ']), autoreset=False,
                     keep_f64_scans=True)
    noise = oracle.noise_table(12345, T + 2, num_beams=nb)
    ors = [oracle.Env(sc, A, noise=noise, ego_idx=cfg['ego_idx'],
                      integrator=oracle.RK4 if cfg['integ'] == 'RK4' else oracle.EULER) for _ in range(B)]
    env.reset(poses)
    oo = [ors[b].reset(poses[b]) for b in range(B)]
    for k in range(T):
        act = np.stack([rng.uniform(-0.4, 0.4, (B, A)), rng.uniform(0, 6, (B, A))], axis=2)
        obs, _, done, info = env.step(act)
        oo = [ors[b].step(act[b]) for b in range(B)]
        st = _np(env.state)
        for b in range(B):
            assert np.allclose(st[b], oo[b]['state'], rtol=0, atol=1e-9), (k, b)
            assert np.allclose(_np(obs['scans_f64'])[b], oo[b]['scans'], rtol=0, atol=1e-9), (k, b)
            assert np.array_equal(_np(obs['collisions'])[b].astype(np.float64), oo[b]['collisions']), (k, b)
            assert np.array_equal(_np(info['collision_idx'])[b].astype(np.float64), oo[b]['collision_idx']), (k, b)
            assert bool(_np(done)[b]) == oo[b]['done'], (k, b)
            assert np.array_equal(_np(info['toggles'])[b].astype(np.float64), oo[b]['toggles']), (k, b)
    env.close()


def test_checkpoint_resume(assets):
    """state_dict()/load_state_dict(): a rollout resumed from a checkpoint (in a fresh env)
    continues bit-identically."""
    import torch
    from red_gym_amd import workload
    B, A = 64, 2
    poses = workload.spawn_poses(B, A)
    acts = torch.as_tensor(workload.action_pool(30, B, A))
    e1 = _vec(assets, B, A, autoreset=True)
    e1.reset(poses)
    for k in range(12):
        e1.step(acts[k])
    ck = e1.state_dict()
    for k in range(12, 30):
        obs1, _, d1, _ = e1.step(acts[k])
    e2 = _vec(assets, B, A, autoreset=True)
    e2.load_state_dict(ck)
    for k in range(12, 30):
        obs2, _, d2, _ = e2.step(acts[k])
    assert torch.equal(e1.state, e2.state) and torch.equal(obs1['scans'], obs2['scans']) and torch.equal(d1, d2)
    assert torch.equal(e1.eng.t['toggles'], e2.eng.t['toggles']) and torch.equal(e1.eng.t['lap_times'], e2.eng.t['lap_times'])
    e1.close(); e2.close()


def test_env_closed_loop_two_agents_golden(golden, assets):
    """Two cars, both on the reference planner's recorded actions, until all(toggles >= 4) (g9, generated by the
    reference's F110Env): _check_done with A > 1 -- the ego's start rotation applied to every car
    (f110_env.py:219-221), lap counts, lap times frozen at 4 toggles while the other car keeps driving, done only
    through all() -- every flag `==` at every step."""
    from red_gym_amd import F110Env, Integrator
    g = golden('g9_env2.npz')
    env = F110Env(map=os.path.join(assets, 'example_map'), map_ext='.png', num_agents=2, timestep=0.01,
                  integrator=Integrator.RK4)
    obs, r, done, info = env.reset(g['start'])
    assert np.allclose(np.array([obs['poses_x'], obs['poses_y'], obs['poses_theta']]).T, g['reset_obs'], atol=1e-12)
    T = g['actions'].shape[0]
    scan_at = dict(zip(g['scan_steps'].tolist(), g['scans']))
    for k in range(T):
        obs, r, done, info = env.step(g['actions'][k])
        assert np.allclose(obs['poses_x'], g['x'][k], rtol=0, atol=1e-7) and np.allclose(obs['poses_y'], g['y'][k], rtol=0, atol=1e-7), k
        assert np.allclose(obs['poses_theta'], g['theta'][k], rtol=0, atol=1e-7), k
        assert np.array_equal(obs['collisions'], g['col'][k]), k
        assert np.array_equal(env.toggle_list, g['toggle'][k]), k
        assert np.array_equal(obs['lap_counts'], g['lap_c'][k]) and np.array_equal(obs['lap_times'], g['lap_t'][k]), k
        assert done == bool(g['done'][k]) and np.array_equal(info['checkpoint_done'], g['ckpt'][k]), k
        if k in scan_at:
            assert np.allclose(np.stack(obs['scans']), scan_at[k], rtol=0, atol=1e-7), k
    tg = g['toggle']
    # the run really exercises what it is for: both cars complete two laps, at different times, without the ego
    # colliding, and the first finisher's lap time stays frozen while its toggles go on
    assert (tg[-1] >= 4).all() and bool(g['done'][-1]) and not g['done'][:-1].any()
    first = int(np.argmax((tg >= 4).any(axis=1)))
    assert first < T - 50 and (g['lap_t'][first:, np.argmax(tg[first] >= 4)] == g['lap_t'][first, np.argmax(tg[first] >= 4)]).all()
    env.close()


def _check_done_replay(x, y, theta, start, col, time, ego_idx=0):
    """Feeds a recorded pose sequence [T,A] step by step to f110_check_done (state carried on the device)."""
    import torch
    from red_gym_amd.engine import check_done
    T, A = x.shape
    dev = torch.device('cuda', 0)
    poses = torch.as_tensor(np.stack([x, y, theta], axis=-1).reshape(T, 1, A, 3), device=dev).contiguous()
    st = torch.as_tensor(np.ascontiguousarray(start, dtype=np.float64).reshape(1, A, 3), device=dev)
    th = -start[ego_idx][2]   # f110_env.py:329, numpy like the reference
    rot = torch.as_tensor(np.array([[np.cos(th), -np.sin(th), np.sin(th), np.cos(th)]]), device=dev)
    cols = torch.as_tensor(np.ascontiguousarray(col, dtype=np.uint8).reshape(T, 1, A), device=dev)
    times = torch.as_tensor(np.ascontiguousarray(time, dtype=np.float64).reshape(T, 1), device=dev)
    near = torch.ones((1, A), dtype=torch.uint8, device=dev)
    tog = torch.zeros((1, A), dtype=torch.int32, device=dev)
    lapt = torch.zeros((1, A), dtype=torch.float64, device=dev)
    out = {k: [] for k in ('toggle', 'lap_c', 'lap_t', 'done', 'ckpt', 'near')}
    for k in range(T):
        lc, dn, ck = check_done(poses[k], st, rot, times[k], cols[k], near, tog, lapt, ego_idx)
        for key, t in (('toggle', tog), ('lap_c', lc), ('lap_t', lapt), ('done', dn), ('ckpt', ck), ('near', near)):
            out[key].append(t.clone())
    return {k: torch.stack(v).cpu().numpy().reshape(T, -1) for k, v in out.items()}


def test_check_done_function_level_golden(golden):
    """f110_check_done (f110_env.py:202-244) on the reference's recorded poses: the 1-agent 2-lap run (g8) and
    the 2-agent run to all(toggles >= 4) (g9); toggles, lap counts, lap times, done, checkpoint_done `==`."""
    g = golden('g8_env.npz')
    T = g['x'].shape[0]
    times = np.cumsum(np.full(T + 1, 0.01))[1:]  # current_time after reset's own step and k+1 steps (:293), same additions
    r = _check_done_replay(g['x'][:, None], g['y'][:, None], g['theta'][:, None], g['start'], g['col'][:, None], times)
    assert np.array_equal(r['toggle'][:, 0], g['toggle']) and np.array_equal(r['lap_c'][:, 0], g['lap_c'])
    assert np.array_equal(r['lap_t'][:, 0], g['lap_t']) and np.array_equal(r['done'][:, 0], g['done'].astype(bool))
    g = golden('g9_env2.npz')
    r = _check_done_replay(g['x'], g['y'], g['theta'], g['start'], g['col'], g['time'])
    assert np.array_equal(r['toggle'], g['toggle']) and np.array_equal(r['lap_c'], g['lap_c'])
    assert np.array_equal(r['lap_t'], g['lap_t']) and np.array_equal(r['done'][:, 0], g['done'].astype(bool))
    assert np.array_equal(r['ckpt'], g['ckpt']) and np.array_equal(r['near'].astype(bool), g['near'])
    # a different ego: the rotation and the collision entry of car 1 are used (done fires when car 1 collides)
    col = np.zeros_like(g['col']); col[5, 1] = 1
    r1 = _check_done_replay(g['x'][:20], g['y'][:20], g['theta'][:20], g['start'], col[:20], g['time'][:20], ego_idx=1)
    assert r1['done'][:, 0].tolist() == [k == 5 for k in range(20)]


def test_check_done_argument_errors():
    import torch
    from red_gym_amd.engine import check_done
    dev = torch.device('cuda', 0)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
    args = [z((2, 2, 3), torch.float64), z((2, 2, 3), torch.float64), z((2, 4), torch.float64), z((2,), torch.float64),
            z((2, 2), torch.uint8), z((2, 2), torch.uint8), z((2, 2), torch.int32), z((2, 2), torch.float64)]
    with pytest.raises(IndexError):
        check_done(*args, ego_idx=2)
    with pytest.raises(ValueError):
        check_done(*(args[:6] + [z((2, 2), torch.int64), args[7]]))


@pytest.mark.parametrize('spec', ['*:0', '*:1', '*:3', '8:0,*:2', '6:2,*:0,10:1', '20:3,*:0,4:2'])
@pytest.mark.parametrize('A', [1, 2])
def test_scan_stage_lists_give_identical_results(assets, spec, A):
    """The wave -> car mapping of a scan launch (one wavefront per car, or 2 / 4 / 8 per car splitting its beam queue,
    in any sequence of stages) must not change a bit: scans, lookup counts per car, state,
    collisions, lap flags of 40 autoreset steps `==` the default mapping (itself checked against the oracle above),
    including cars that start inside a wall (no ray leaves the car), masked resets (scan launches that skip cars)
    and the function-level scan with poses off the map."""
    import torch
    from red_gym_amd import workload
    B, T = 50, 40
    poses = workload.spawn_poses(B, A)
    poses[3, 0, :2] = [0.0, 20.0]       # inside a wall / off the track: d0 <= eps or > max_range paths
    poses[17, A - 1, :2] = [-78.0, -44.0]
    poses[18, 0, :2] = [500.0, 500.0]   # off the map: dt[-1,-1] = 51 m > max_range
    acts = torch.as_tensor(workload.action_pool(8, B, A), device='cuda')
    e1 = _vec(assets, B, A, autoreset=True, count_lookups=True)
    e2 = _vec(assets, B, A, autoreset=True, count_lookups=True)
    e2.eng.set_scan_stages(spec)
    e1.reset(poses); e2.reset(poses)
    keys = ('scans_f64', 'scans', 'state', 'lookups', 'collisions', 'in_collision', 'toggles', 'done', 'noise_step')
    for k in range(T):
        e1.step(acts[k % 8]); e2.step(acts[k % 8])
        if k == 20:   # masked reset: only some envs take part in the reset's scan launch
            m = torch.zeros(B, dtype=torch.uint8, device='cuda'); m[::3] = 1
            e1.reset(poses, m); e2.reset(poses, m)
        for key in keys:
            assert torch.equal(e1.eng.t[key], e2.eng.t[key]), (k, key)
    assert int(e1.eng.t['done'].sum()) >= 0 and int(e1.eng.t['lookups'].min()) > 0
    # function-level scan (pose stride 3, no noise), odd pose count
    rng = np.random.default_rng(5)
    ps = np.concatenate([workload.spawn_poses(37, 1)[:, 0], rng.uniform(-120, 120, (6, 3))])
    a64, a32, alk = e1.eng.scan(ps, want_f32=True, want_lookups=True)
    b64, b32, blk = e2.eng.scan(ps, want_f32=True, want_lookups=True)
    assert torch.equal(a64, b64) and torch.equal(a32, b32) and torch.equal(alk, blk)
    e1.close(); e2.close()


def test_twelve_agents_vs_oracle(assets):
    """The reference takes any num_agents (f110_env.py:131-134, base_classes.py:484-490); the handle takes up to
    F110_MAX_AGENTS = 32.  12 cars per env in a queue on the raceline, 1 m apart (neighbours overlap in their scans and
    some touch): 40 random steps against oracle envs -- state / scans 1e-9, collisions, collision_idx (last partner in
    pair order), toggles and done `==`."""
    B, A, T = 3, 12, 40
    rl = np.loadtxt(os.path.join(assets, 'example_waypoints.csv'), delimiter=';', skiprows=3)
    rng = np.random.default_rng(12)
    poses = np.zeros((B, A, 3))
    for b in range(B):
        for a in range(A):
            k = (200 * b + 300 - 5 * a) % rl.shape[0]
            poses[b, a] = [rl[k, 1] + rng.normal(0, 0.05), rl[k, 2] + rng.normal(0, 0.05), rl[k, 3] + np.pi / 2]
    poses[1, 5, :2] = poses[1, 4, :2] + 0.2   # two cars on top of each other: GJK hit from the start
    env = _vec(assets, B, A, ego_idx=7)
    ors = [oracle.Env(_mk_scanner(assets), A, noise=oracle.noise_table(12345, T + 2), ego_idx=7) for _ in range(B)]
    env.reset(poses)
    oo = [ors[b].reset(poses[b]) for b in range(B)]
    hits = 0
    for k in range(T):
        act = np.stack([rng.uniform(-0.4189, 0.4189, (B, A)), rng.uniform(0, 6, (B, A))], axis=2)
        obs, _, done, info = env.step(act)
        oo = [ors[b].step(act[b]) for b in range(B)]
        st = _np(env.state)
        for b in range(B):
            assert np.allclose(st[b], oo[b]['state'], rtol=0, atol=1e-9), (k, b)
            assert np.allclose(_np(obs['scans_f64'])[b], oo[b]['scans'], rtol=0, atol=1e-9), (k, b)
            assert np.array_equal(_np(obs['collisions'])[b].astype(np.float64), oo[b]['collisions']), (k, b)
            assert np.array_equal(_np(info['collision_idx'])[b].astype(np.float64), oo[b]['collision_idx']), (k, b)
            assert np.array_equal(_np(info['toggles'])[b].astype(np.float64), oo[b]['toggles']), (k, b)
            assert bool(_np(done)[b]) == oo[b]['done'], (k, b)
        hits += int(_np(obs['collisions']).sum())
    assert hits > 0
    from red_gym_amd import F110VecEnv
    with pytest.raises(ValueError):
        F110VecEnv(1, map=os.path.join(assets, 'example_map'), num_agents=33)
    env.close()


def test_set_scan_stages_refuses_malformed_lists(assets):
    env = _vec(assets, 8, 1)
    for bad in ('x', '4', '*:9', '*:0,*:1', '4:0:1', '1:0,1:0,1:0,1:0,1:0,1:0,*:0', '100:0,*:1', '*:-1', '2:0,'):
        if bad == '4':
            continue  # "4 cars, whole waves, the rest whole": legal
        with pytest.raises(ValueError):
            env.eng.set_scan_stages(bad)
    env.eng.set_scan_stages('4:1,*:0')
    env.eng.set_scan_stages(None)
    env.close()


@pytest.mark.parametrize('seed', list(range(32)))
def test_step_path_fuzz_vs_oracle(assets, seed):
    """Random everything the step path is parameterised by -- map (size, resolution incl. non-powers of two, origin,
    origin yaw, obstacles), vehicle parameters, fov, beam count, agents, ego index, integrator, time step -- 25 random
    steps of 5 envs against independent oracle envs: state / scans 1e-9, every flag, index and lap toggle `==`."""
    import torch
    from scipy.ndimage import distance_transform_edt
    from red_gym_amd import F110VecEnv, Integrator
    from red_gym_amd.engine import DEFAULT_PARAMS
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(90, 320)), int(rng.integers(90, 320))
    res = float(rng.choice([0.03125, 0.05, 0.0625, 0.08, 0.1, 0.125]))
    free = np.ones((H, W), np.uint8)
    free[[0, -1], :] = 0; free[:, [0, -1]] = 0
    for _ in range(int(rng.integers(3, 12))):                      # rectangles and discs
        r0, c0 = int(rng.integers(0, H)), int(rng.integers(0, W))
        if rng.random() < 0.5:
            free[r0:r0 + int(rng.integers(1, 25)), c0:c0 + int(rng.integers(1, 25))] = 0
        else:
            rr, cc = np.ogrid[:H, :W]
            free[(rr - r0) ** 2 + (cc - c0) ** 2 <= int(rng.integers(1, 15)) ** 2] = 0
    if seed % 4 == 3:
        free[0, :] = 1                                              # an open side: rays leave the map (dt[-1,-1] read)
    ox, oy = float(rng.uniform(-30, 10)), float(rng.uniform(-30, 10))
    oth = float(rng.uniform(-3, 3)) if seed % 3 == 0 else 0.0       # rotated origin: the general index path
    dt = res * distance_transform_edt(free)
    m = {'height': H, 'width': W, 'resolution': res, 'orig_x': ox, 'orig_y': oy, 'orig_s': float(np.sin(oth)),
         'orig_c': float(np.cos(oth)), 'dt': np.ascontiguousarray(dt), 'img': free * 255.}
    nb = int(rng.choice([64, 100, 271, 700, 1080]))
    fov = float(rng.uniform(2.0, 2 * np.pi))
    A = int(rng.integers(1, 4))
    ego = int(rng.integers(0, A))
    integ = 'RK4' if rng.random() < 0.6 else 'Euler'
    tstep = float(rng.choice([0.01, 0.02, 0.005]))
    params = dict(DEFAULT_PARAMS)
    for k in ('mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 'a_max', 'v_max', 'width', 'length', 'sv_max', 's_max'):
        params[k] = params[k] * float(rng.uniform(0.8, 1.25))
    params['s_min'], params['sv_min'] = -params['s_max'], -params['sv_max']
    B, T = 5, 25
    sc = oracle.Scanner(nb, fov, params=params)
    sc.set_map_dict(m)
    clear = np.argwhere(dt > 0.35)
    pick = clear[rng.choice(len(clear), size=B * A, replace=len(clear) < B * A)]
    # cell centres -> world (rotate by the origin yaw, then translate: the inverse of laser_models.py:71-77)
    xr, yr = (pick[:, 1] + 0.5) * res, (pick[:, 0] + 0.5) * res
    poses = np.zeros((B, A, 3))
    poses[..., 0] = (ox + np.cos(oth) * xr - np.sin(oth) * yr).reshape(B, A)
    poses[..., 1] = (oy + np.sin(oth) * xr + np.cos(oth) * yr).reshape(B, A)
    poses[..., 2] = rng.uniform(-1, 7, (B, A))
    if A > 1:
        poses[0, 1, :2] = poses[0, 0, :2] + 0.25                    # a car on top of another: GJK, opponent cast
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), map_ext='.png', num_agents=A, fov=fov, num_beams=nb,
                     ego_idx=ego, integrator=getattr(Integrator, integ), autoreset=False, keep_f64_scans=True,
                     params=params, timestep=tstep)
    env.update_map_occupancy(free, res, ox, oy, oth)
    noise = oracle.noise_table(12345, T + 2, num_beams=nb)
    ors = [oracle.Env(sc, A, params=params, time_step=tstep, noise=noise, ego_idx=ego,
                      integrator=oracle.RK4 if integ == 'RK4' else oracle.EULER) for _ in range(B)]
    env.reset(poses)
    oo = [ors[b].reset(poses[b]) for b in range(B)]
    tag = dict(seed=seed, H=H, W=W, res=res, oth=oth, nb=nb, A=A, integ=integ)
    sane = [True] * B
    for k in range(T):
        act = np.stack([rng.uniform(-0.45, 0.45, (B, A)), rng.uniform(-1, 7, (B, A))], axis=2)
        obs, _, done, info = env.step(act)
        oo = [ors[b].step(act[b]) for b in range(B)]
        st, s64 = _np(env.state), _np(obs['scans_f64'])
        col, cix, dn, tg = _np(obs['collisions']), _np(info['collision_idx']), _np(done), _np(info['toggles'])
        for b in range(B):
            # Random vehicle parameters can make the single-track model blow up (yaw 1e9 rad, yaw rate 1e11 within a
            # few steps).  Such an env is dropped from the comparison from then on: the opponent ray cast adds the beam
            # angle by rotating table entries where the reference rounds yaw + angle to fp64 first, which is the same to
            # 1e-15 for any physical yaw but differs by the spacing of doubles at yaw = 1e9 (2e-7 rad).
            sane[b] = sane[b] and bool(np.abs(oo[b]['state']).max() < 1e4)
            if not sane[b]:
                continue
            assert np.allclose(st[b], oo[b]['state'], rtol=0, atol=1e-9), (k, b, tag)
            if not np.allclose(s64[b], oo[b]['scans'], rtol=0, atol=1e-9):
                dlt = np.abs(s64[b] - oo[b]['scans'])
                a, i = np.unravel_index(np.argmax(dlt), dlt.shape)
                raise AssertionError('scan mismatch step %d env %d car %d beam %d: gpu %.17g oracle %.17g (%d beams differ); '
                                     'state %s; %s' % (k, b, a, i, s64[b][a, i], oo[b]['scans'][a, i], int((dlt > 1e-9).sum()),
                                                       st[b][a].tolist(), tag))
            assert np.array_equal(col[b].astype(np.float64), oo[b]['collisions']), (k, b, tag)
            assert np.array_equal(cix[b].astype(np.float64), oo[b]['collision_idx']), (k, b, tag)
            assert bool(dn[b]) == oo[b]['done'] and np.array_equal(tg[b].astype(np.float64), oo[b]['toggles']), (k, b, tag)
    env.close()


def test_plain_store_instantiation_matches_streaming(assets, tmp_path):
    """Launches of more than 327 680 cars run scan_kernel<.., 2> (ordinary instead of streaming stores for the fp32 scan,
    profiles/r04_scan_stores.txt L), a size no test reaches: F110_SCAN_STORES=plain forces that instantiation (the variable is read
    once per process, hence the child processes).  Same seeds, same actions: every observation of 40 steps is identical."""
    import subprocess
    import sys
    script = (
        "import sys, os, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from red_gym_amd import F110VecEnv, workload\n"
        "env = F110VecEnv(96, map=%r, num_agents=2, autoreset=True)\n"
        "env.reset(workload.spawn_poses(96, 2))\n"
        "acts = torch.as_tensor(workload.action_pool(4, 96, 2), device='cuda')\n"
        "out = []\n"
        "for k in range(40):\n"
        "    obs = env.step(acts[k %% 4])[0]\n"
        "    out.append(np.concatenate([obs['scans'].double().cpu().numpy().ravel(), obs['poses_x'].cpu().numpy().ravel(), obs['collisions'].double().cpu().numpy().ravel()]))\n"
        "np.save(sys.argv[1], np.stack(out))\n"
        "env.close()\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(assets, 'example_map')))
    res = {}
    for mode in ('stream', 'plain'):
        path = str(tmp_path / ('obs_%s.npy' % mode))
        env = dict(os.environ, F110_SCAN_STORES=mode)
        r = subprocess.run([sys.executable, '-c', script, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = np.load(path)
    assert res['stream'].shape == res['plain'].shape and np.array_equal(res['stream'], res['plain'])


@pytest.mark.parametrize('kind', ['wide', 'mixed_sign', 'negative', 'non_finite'])
def test_ittc_with_custom_side_tables_vs_oracle(assets, kind):
    """The scan reads a beam's side distance only where the iTTC test can fire (scan value below the largest finite side
    distance + the candidate margin).  Side tables the default vehicle never produces -- wide, of mixed sign, all negative,
    with inf / NaN entries -- installed through f110_set_tables: collisions, states and scans of a wall-hugging rollout
    stay identical to oracle envs holding the same table (check_ttc_jit, laser_models.py:189-217)."""
    from red_gym_amd import F110VecEnv, _lib
    from red_gym_amd.engine import _np_ptr
    B, T, nb = 8, 60, 1080
    sc = oracle.Scanner(nb, 2 * np.pi)
    sc.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    rng = np.random.default_rng(5)
    side = np.array(sc.side_distances)
    if kind == 'wide':
        side = rng.uniform(0.2, 1.5, nb)
    elif kind == 'mixed_sign':
        side = rng.uniform(-0.5, 0.8, nb)
    elif kind == 'negative':
        side = -rng.uniform(0.01, 0.6, nb)
    else:
        side = rng.uniform(0.1, 0.9, nb)
        side[::7] = np.inf; side[3::11] = -np.inf; side[5::13] = np.nan
    sc.side_distances[:] = side
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, autoreset=False, keep_f64_scans=True)
    e = env.eng
    e.side_distances = np.ascontiguousarray(side)
    _lib.check(e.lib.f110_set_tables(e._h, _np_ptr(e.sines), _np_ptr(e.cosines), _np_ptr(e.scan_angles), _np_ptr(e.beam_cosines),
                                     _np_ptr(e.side_distances)))
    m = sc.map
    near = np.argwhere((m['dt'] > 0.25) & (m['dt'] < 0.9))    # close to walls: many scan values within reach of the side table
    pick = near[rng.choice(len(near), size=B, replace=False)]
    poses = np.zeros((B, 1, 3))
    poses[:, 0, 0] = m['orig_x'] + (pick[:, 1] + 0.5) * m['resolution']
    poses[:, 0, 1] = m['orig_y'] + (pick[:, 0] + 0.5) * m['resolution']
    poses[:, 0, 2] = rng.uniform(0, 6.28, B)
    noise = oracle.noise_table(12345, T + 2, num_beams=nb)
    ors = [oracle.Env(sc, 1, noise=noise) for _ in range(B)]
    env.reset(poses)
    for b in range(B):
        ors[b].reset(poses[b])
    hits = 0
    for k in range(T):
        act = np.stack([rng.uniform(-0.4, 0.4, (B, 1)), rng.uniform(0.5, 7, (B, 1))], axis=2)
        obs, _, done, info = env.step(act)
        oo = [ors[b].step(act[b]) for b in range(B)]
        st = _np(env.state)
        for b in range(B):
            assert np.array_equal(_np(obs['collisions'])[b].astype(np.float64), oo[b]['collisions']), (kind, k, b)
            assert np.allclose(st[b], oo[b]['state'], rtol=0, atol=1e-9), (kind, k, b)
            assert np.allclose(_np(obs['scans_f64'])[b], oo[b]['scans'], rtol=0, atol=1e-9), (kind, k, b)
            hits += int(oo[b]['collisions'][0])
    if kind in ('wide', 'mixed_sign'):
        assert hits > 0   # the rare path did fire
    env.close()
