"""The reference's Python surface on top of the HIP path: gym.make('f110_gym:f110-v0'),
the waypoint-follow caller (config 1), ScanSimulator2D and Simulator mirrors."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_waypoint_follow_example_config1():
    """examples/waypoint_follow.py: gym.make -> reset -> plan/step loop until done.
    Reference result: 3329 steps, 33.29 s sim time, 2 laps, no collision."""
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    import waypoint_follow
    steps, laptime, obs = waypoint_follow.main()
    assert steps == 3329 and abs(laptime - 33.29) < 1e-9
    assert obs['lap_counts'][0] == 2 and obs['collisions'][0] == 0


def test_lidar_example_images_match_oracle():
    """examples/lidar_example.py = the reference's examples/lidar_example.py:76-107 without its windows: the two images it draws
    from every ego scan (RAYS, 50 beams, black, 3 channels, fov 4.7; FILL, white, 3 channels) through the same
    `from weap_util.lidar import lidar_to_bitmap` call, against the oracle's rasteriser on the scans the loop produced."""
    from oracle import bitmap as ob
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    import lidar_example
    scans, blinded, nonblinded, done = lidar_example.main(steps=120, keep=10)
    assert not done and scans.shape == (12, 1080) and blinded.shape == nonblinded.shape == (12, 256, 256, 3)
    assert blinded.dtype == nonblinded.dtype == np.uint8
    want_b = ob.lidar_to_bitmap(scans, channels=3, fov=lidar_example.FOV, target_beam_count=50, draw_mode='RAYS', bg_color='black')
    want_f = ob.lidar_to_bitmap(scans, channels=3, fov=lidar_example.FOV, draw_mode='FILL', bg_color='white')
    assert np.array_equal(blinded, want_b) and np.array_equal(nonblinded, want_f)
    assert (nonblinded != 255).any() and (blinded != 0).any()   # something was drawn


def test_scan_simulator_mirror(assets, golden):
    from f110_gym.envs.laser_models import ScanSimulator2D
    g = golden('g1_scan.npz')
    sim = ScanSimulator2D(1080, 2 * np.pi)
    with pytest.raises(ValueError, match='Map is not set'):
        sim.scan(np.zeros(3), None)
    assert sim.set_map(os.path.join(assets, 'example_map.yaml'), '.png') is True
    assert sim.get_increment() == 2 * np.pi / 1079
    for k in (0, 9, 41):
        assert np.array_equal(sim.scan(g['ex_poses'][k], None), g['ex_scans'][k])
    # laser_models.py:554-580 test_rng: same seed -> same noisy scan; consecutive scans differ
    rng1, rng2 = np.random.default_rng(seed=12345), np.random.default_rng(seed=12345)
    s1, s2 = sim.scan(g['ex_poses'][0], rng1), sim.scan(g['ex_poses'][0], rng2)
    assert np.array_equal(s1, s2) and not np.array_equal(s1, sim.scan(g['ex_poses'][0], rng1))
    gn = golden('g2_noise.npz')['seed12345'][0]
    assert np.array_equal(s1, g['ex_scans'][0] + gn)


def test_simulator_mirror_matches_golden(assets, golden):
    from f110_gym.envs.base_classes import Integrator, Simulator
    g = golden('g7_sim.npz')
    sim = Simulator(oracle.DEFAULT_PARAMS, 2, 12345, 2 * np.pi, time_step=0.01, integrator=Integrator.RK4)
    with pytest.raises(ValueError):
        sim.reset(g['a2_start'])
    sim.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    with pytest.raises(ValueError, match='Number of poses'):
        sim.reset(np.zeros((3, 3)))
    with pytest.raises(IndexError):
        sim.update_params(oracle.DEFAULT_PARAMS, agent_idx=2)
    sim.reset(g['a2_start'])
    obs = sim.step(np.zeros((2, 2)))
    assert set(obs) == {'ego_idx', 'scans', 'poses_x', 'poses_y', 'poses_theta', 'linear_vels_x', 'linear_vels_y',
                        'ang_vels_z', 'collisions'}
    for k in range(120):
        obs = sim.step(g['a2_actions'][k])
        assert np.allclose(sim.agents[0].state, g['a2_states'][k, 0], rtol=0, atol=1e-9)
        assert np.allclose(sim.agents[1].state, g['a2_states'][k, 1], rtol=0, atol=1e-9)
        assert np.array_equal(obs['collisions'], g['a2_collisions'][k])
        assert np.array_equal(sim.collision_idx, g['a2_collision_idx'][k])
    assert g['a2_collisions'][:120].any()


def test_f110env_kwargs_and_errors(assets):
    from red_gym_amd import compat
    compat.install_missing()
    import gym
    env = gym.make('f110_gym:f110-v0', map=os.path.join(assets, 'example_map'), map_ext='.png', num_agents=2,
                   render_options={'ignored': True})
    assert env.num_agents == 2 and env.timestep == 0.01 and env.ego_idx == 0 and env.seed == 12345
    with pytest.raises(ValueError, match='Number of poses'):
        env.reset(np.zeros((1, 3)))
    obs, r, done, info = env.reset(np.array([[0.7, 0.0, 1.37], [0.7, -1.5, 1.37]]))
    assert len(obs['scans']) == 2 and obs['collisions'].shape == (2,) and info['checkpoint_done'].shape == (2,)
    assert obs['linear_vels_y'] == [0., 0.] and r == 0.01
    env.render()  # accepted, no-op
    env.close()


def test_hip_pure_pursuit_matches_numpy_planner(assets):
    """batched pure pursuit on the GPU vs the NumPy restatement of
    examples/waypoint_follow.py:15-217 (which reproduces the reference's recorded actions)."""
    from argparse import Namespace
    import yaml
    from red_gym_amd import F110VecEnv, workload
    from oracle.planner import PurePursuitPlanner
    conf = Namespace(**yaml.safe_load(open(os.path.join(assets, 'config_example_map.yaml'))))
    conf.wpt_path = os.path.join(assets, 'example_waypoints.csv')
    pl = PurePursuitPlanner(conf, 0.17145 + 0.15875)
    wp = np.stack([pl.waypoints[:, conf.wpt_xind], pl.waypoints[:, conf.wpt_yind], pl.waypoints[:, conf.wpt_vind]], axis=1)
    B = 512
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False)
    rng = np.random.default_rng(4)
    poses = workload.spawn_poses(B, 1)
    poses[:64, 0, :2] += rng.normal(0, 1.0, (64, 2))     # off the raceline (> lookahead: re-acquire branch)
    poses[64:72, 0, :2] += 100.0                         # farther than max_reacquire: (4.0, 0.0)
    poses[72:80, 0, :] = [wp[-1, 0], wp[-1, 1], 1.0]     # at the last waypoint: wrap-around search
    env.reset(poses)
    tlad, vgain = 0.82461887897713965, 1.375
    act = env.pure_pursuit(wp, tlad, vgain).cpu().numpy()
    st = env.state.cpu().numpy()
    for b in range(B):
        sp, stg = pl.plan(st[b, 0, 0], st[b, 0, 1], st[b, 0, 4], tlad, vgain)
        assert abs(act[b, 0, 0] - stg) < 1e-12 and abs(act[b, 0, 1] - sp) < 1e-12, b
    assert (act[64:72, 0, 1] == 4.0).all() and (act[64:72, 0, 0] == 0.0).all()
    # closed loop on the GPU: 16 cars race 400 steps without touching the host planner
    env2 = F110VecEnv(16, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False, keep_f64_scans=True)
    p2 = np.repeat(np.array([[[conf.sx, conf.sy, conf.stheta]]]), 16, axis=0)
    env2.reset(p2)
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(workload.EXAMPLE_MAP + '.yaml', '.png')
    oenv = oracle.Env(sc, 1, noise=oracle.noise_table(12345, 402))
    oo = oenv.reset(p2[0])
    for k in range(400):
        env2.step(env2.pure_pursuit(wp, tlad, vgain))
        sp, stg = pl.plan(oo['state'][0, 0], oo['state'][0, 1], oo['state'][0, 4], tlad, vgain)
        oo = oenv.step(np.array([[stg, sp]]))
    s = env2.state.cpu().numpy()
    assert np.allclose(s[0, 0], oo['state'][0], rtol=0, atol=1e-7) and np.array_equal(s[0], s[15])
    assert s[0, 0, 3] > 5.0  # racing, not crashed
    env.close(); env2.close()


def test_trajectory_recorder(assets, tmp_path):
    """Headless replacement of the render window: records poses / laps (and scans) on the
    device and dumps an .npz that replays the run."""
    from red_gym_amd import F110Env, F110VecEnv, workload
    from red_gym_amd.recorder import TrajectoryRecorder
    env = F110VecEnv(32, map=workload.EXAMPLE_MAP, num_agents=2, autoreset=True)
    rec = TrajectoryRecorder(env, max_steps=20, envs=[3, 7, 31], with_scans=True)
    poses, acts = workload.spawn_poses(32, 2), workload.action_pool(20, 32, 2)
    env.reset(poses)
    rec.record()
    for k in range(19):
        env.step(acts[k])
        rec.record()
    with pytest.raises(IndexError):
        rec.record()
    path = rec.save(str(tmp_path / 'traj.npz'))
    d = np.load(path)
    assert d['state'].shape == (20, 3, 2, 7) and d['scans'].shape == (20, 3, 2, 1080) and list(d['env_index']) == [3, 7, 31]
    assert np.array_equal(d['state'][-1], env.state[[3, 7, 31]].cpu().numpy())
    assert np.allclose(d['state'][0, :, :, 0:2], poses[[3, 7, 31], :, 0:2])
    env.close()
    e1 = F110Env(map=os.path.join(assets, 'example_map'), num_agents=1)
    e1.start_recording(5)
    obs, *_ = e1.reset(np.array([[0.7, 0.0, 1.37]]))
    e1.render()
    for _ in range(3):
        obs, *_ = e1.step(np.array([[0.0, 1.0]]))
        e1.render(mode='human_fast')
    out = np.load(e1.save_recording(str(tmp_path / 'single.npz')))
    assert out['state'].shape == (4, 1, 1, 7) and abs(out['state'][-1, 0, 0, 0] - obs['poses_x'][0]) < 1e-15
    e1.close()


def test_lidar_dataset_writer_reference_format(golden, assets, tmp_path):
    """f4: LidarDatasetWriter writes the reference's dataset format (f1tenth_gym/examples/lidar.py:250-254: key
    'data', uint8 [N,256,256] of 0/1) and, fed the scans the reference's own main() saw (g12), the reference's own
    array; record_episodes runs the reference's recording loop on a batch."""
    import torch
    from red_gym_amd import F110VecEnv
    from red_gym_amd.recorder import LidarDatasetWriter
    g = golden('g12_pointgrid.npz')
    shape = tuple(g['shape'])
    want = np.unpackbits(g['data_bits'], axis=-1)[..., :shape[-1]].reshape(shape)
    w = LidarDatasetWriter('cuda:0', capacity=64)
    scans = torch.as_tensor(g['scans'], device='cuda')
    assert w.add(scans[:20]) == 20 and w.add(scans[20:]) == shape[0] - 20
    path = str(tmp_path / 'lidar_dataset_ep5.npz')
    assert w.save(path) == shape[0]
    f = np.load(path, allow_pickle=False)
    assert f.files == ['data'] and f['data'].dtype == np.uint8 and f['data'].shape == shape
    assert np.array_equal(f['data'], want)
    with pytest.raises(IndexError):
        w.add(torch.zeros((64, 1080), device='cuda'))
    # the recording loop on a batch: 32 episodes x <= 10 steps, frames only while the episode is alive
    env = F110VecEnv(32, map=os.path.join(assets, 'example_map'), map_ext='.png', num_agents=1, autoreset=False, fov=4.7)
    ds = LidarDatasetWriter.record_episodes(env, steps=10, seed=3)
    d = ds.array()
    assert d.dtype == np.uint8 and d.shape[1:] == (256, 256) and 32 <= d.shape[0] <= 320 and set(np.unique(d)) <= {0, 1}
    assert d.reshape(len(d), -1).sum(1).min() > 0
    env.close()


def _example_raceline(assets):
    from argparse import Namespace
    import yaml
    conf = Namespace(**yaml.safe_load(open(os.path.join(assets, 'config_example_map.yaml'))))
    w = np.loadtxt(os.path.join(assets, 'example_waypoints.csv'), delimiter=conf.wpt_delim, skiprows=conf.wpt_rowskip)
    return np.ascontiguousarray(w[:, [conf.wpt_xind, conf.wpt_yind, conf.wpt_vind]])


def _hip_plan(wp, poses, lookahead, vgain, wheelbase=0.17145 + 0.15875):
    """f110_pure_pursuit on poses [n,3] = (x, y, theta) -> actions [n,2] = (steer, speed)"""
    import ctypes as C
    import torch
    from red_gym_amd.engine import _lib, _ptr
    dev = torch.device('cuda', 0)
    st = np.zeros((len(poses), 7))
    st[:, [0, 1, 4]] = poses
    st = torch.as_tensor(st, device=dev)
    w = torch.as_tensor(np.ascontiguousarray(wp, dtype=np.float64), device=dev)
    out = torch.empty((len(poses), 2), dtype=torch.float64, device=dev)
    _lib.check(_lib.load().f110_pure_pursuit(None, _ptr(w), w.shape[0], float(lookahead), float(vgain), float(wheelbase), 20.0,
                                             _ptr(st), len(poses), _ptr(out), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out.cpu().numpy()


def test_hip_pure_pursuit_reproduces_reference_recorded_actions(golden, assets):
    """ONE hop from the kernel to the reference: the poses the reference's own PurePursuitPlanner
    (examples/waypoint_follow.py:183-217) saw in its closed loops -- g8: 1 agent, 3 329 steps; g9: 2 agents with their
    own vgain, 3 509 steps -- go through f110_pure_pursuit and must give the actions it recorded: the speed `==` (a
    table value times vgain), the steering angle to 1e-9 (trigonometry of the actuation)."""
    wp = _example_raceline(assets)
    tlad = 0.82461887897713965
    g = golden('g8_env.npz')
    poses = np.concatenate([g['reset_obs'][None, :3], np.stack([g['x'], g['y'], g['theta']], axis=1)[:-1]])
    act = _hip_plan(wp, poses, tlad, 1.375)
    assert np.array_equal(act[:, 1], g['actions'][:, 1])
    assert np.abs(act[:, 0] - g['actions'][:, 0]).max() < 1e-9
    g = golden('g9_env2.npz')
    for a in range(2):
        poses = np.concatenate([g['reset_obs'][a][None], np.stack([g['x'][:, a], g['y'][:, a], g['theta'][:, a]], axis=1)[:-1]])
        act = _hip_plan(wp, poses, tlad, g['vgains'][a])
        assert np.array_equal(act[:, 1], g['actions'][:, a, 1]), a
        assert np.abs(act[:, 0] - g['actions'][:, a, 0]).max() < 1e-9, a


def test_hip_pure_pursuit_prepared_raceline_is_the_same_planner(assets, golden):
    """f110_pure_pursuit_prepare: the grid of candidate lists must not change a single bit of what the planner answers.
    200 000 poses -- on the raceline, scattered around it (every cell of the grid several times), far outside the grid, NaN --
    through the prepared (one lane per car) and the unprepared (one wavefront per car) kernel: actions `==`.  Also the golden
    closed loops of the reference's planner (g8), a raceline whose centre of curvature overflows a cell's list (a circle:
    every segment is equally near its centre), a degenerate raceline, and a 2-point raceline."""
    import ctypes as C
    import torch
    from red_gym_amd import F110VecEnv, workload
    from red_gym_amd.engine import _lib, _ptr
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    env = F110VecEnv(4, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False)   # (a handle to keep the grid in)
    h = env.eng._h
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def plan(w, poses, prepared, tlad=0.82461887897713965, vgain=1.375):
        st = np.zeros((len(poses), 7)); st[:, [0, 1, 4]] = poses
        st = torch.as_tensor(st, device=dev)
        out = torch.empty((len(poses), 2), dtype=torch.float64, device=dev)
        _lib.check(lib.f110_pure_pursuit(h if prepared else None, _ptr(w), w.shape[0], tlad, vgain, 0.17145 + 0.15875, 20.0, _ptr(st), len(poses),
                                         _ptr(out), stream))
        return out.cpu().numpy()

    rng = np.random.default_rng(11)
    wp = _example_raceline(assets)
    circle = np.stack([5 * np.cos(np.linspace(0, 2 * np.pi, 400, endpoint=False)), 5 * np.sin(np.linspace(0, 2 * np.pi, 400, endpoint=False)),
                       np.full(400, 3.0)], axis=1)
    bad = np.concatenate([wp[:200], wp[199:200], wp[200:]])
    two = np.array([[0.0, 0.0, 2.0], [3.0, 1.0, 2.5]])
    for name, line in (('example', wp), ('circle', circle), ('degenerate', bad), ('two points', two)):
        w = torch.as_tensor(np.ascontiguousarray(line), device=dev)
        _lib.check(lib.f110_pure_pursuit_prepare(h, _ptr(w), w.shape[0], 0.0, 0.0, stream))
        n = 200000 if name == 'example' else 20000
        k = rng.integers(0, len(line), n)
        poses = np.stack([line[k, 0], line[k, 1], rng.uniform(-np.pi, np.pi, n)], axis=1)
        poses[:, :2] += rng.normal(0, 1.0, (n, 2)) * rng.choice([0.0, 0.05, 0.4, 1.5, 4.0], (n, 1))
        lo, hi = line[:, :2].min(0) - 3.5, line[:, :2].max(0) + 3.5
        m = n // 10
        poses[:m, :2] = rng.uniform(lo, hi, (m, 2))                       # uniformly over (and just past) the grid
        poses[m:m + 50, :2] = rng.uniform(-500, 500, (50, 2))             # far outside
        poses[m + 50:m + 54, 0] = np.nan                                  # NaN poses
        poses[m + 54:m + 58, :2] = 0.0                                    # the circle's centre: every segment equally near
        a0, a1 = plan(w, poses, False), plan(w, poses, True)
        assert np.array_equal(a0, a1, equal_nan=True), (name, int((a0 != a1).sum()))
        if name == 'degenerate':
            assert (a1 == [0.0, 4.0]).all()
    # the reference's own recorded actions through the prepared kernel
    w = torch.as_tensor(np.ascontiguousarray(wp), device=dev)
    _lib.check(lib.f110_pure_pursuit_prepare(h, _ptr(w), w.shape[0], 0.0, 0.0, stream))
    g = golden('g8_env.npz')
    poses = np.concatenate([g['reset_obs'][None, :3], np.stack([g['x'], g['y'], g['theta']], axis=1)[:-1]])
    act = plan(w, poses, True)
    assert np.array_equal(act[:, 1], g['actions'][:, 1]) and np.abs(act[:, 0] - g['actions'][:, 0]).max() < 1e-9
    # Engine.pure_pursuit prepares a raceline tensor on its second use and plans the same
    wt = torch.as_tensor(np.ascontiguousarray(wp), device=dev)
    env.reset(workload.spawn_poses(4, 1))
    b0 = env.pure_pursuit(wt, 0.8, 1.2).clone()
    b1 = env.pure_pursuit(wt, 0.8, 1.2).clone()
    b2 = env.pure_pursuit(wt, 0.8, 1.2).clone()
    assert env.eng._plan_key is not None and torch.equal(b0, b1) and torch.equal(b1, b2)
    env.close()


def test_hip_pure_pursuit_degenerate_raceline(assets):
    """Two equal consecutive waypoints: the reference's nearest-point search divides 0 by 0 on that segment, np.argmin
    returns it, and the NaN distance makes plan() answer (4.0, 0.0) for every pose (waypoint_follow.py:16-47, :189-212).
    Both planner kernels (LDS, global memory) and the checker do the same."""
    import torch
    from oracle.planner import PurePursuitPlanner, Raceline
    from red_gym_amd.engine import TrackSet, _lib, _ptr
    import ctypes as C
    wp = _example_raceline(assets)
    bad = np.concatenate([wp[:200], wp[199:200], wp[200:]])
    poses = np.stack([wp[::50, 0], wp[::50, 1], np.zeros(len(wp[::50]))], axis=1)
    pl = PurePursuitPlanner.__new__(PurePursuitPlanner)
    pl.wheelbase, pl.max_reacquire, pl.line, pl.speeds = 0.33, 20., Raceline(bad[:, :2]), bad[:, 2]
    with np.errstate(all='ignore'):
        assert pl.plan(poses[3, 0], poses[3, 1], 0.3, 0.9, 1.2) == (4.0, 0.0)
    act = _hip_plan(bad, poses, 0.9, 1.2)
    assert np.array_equal(act, np.tile([0.0, 4.0], (len(poses), 1)))
    assert not np.array_equal(_hip_plan(wp, poses, 0.9, 1.2), act)
    # global-memory form: raceline 0 sound, raceline 1 degenerate
    dev = torch.device('cuda', 0)
    ts = TrackSet([wp, bad], dev)
    st = np.zeros((2 * len(poses), 7)); st[:, [0, 1, 4]] = np.concatenate([poses, poses])
    st = torch.as_tensor(st, device=dev)
    of_car = torch.as_tensor(np.repeat([0, 1], len(poses)).astype(np.int32), device=dev)
    out = torch.empty((2 * len(poses), 2), dtype=torch.float64, device=dev)
    _lib.check(_lib.load().f110_pure_pursuit_tracks(None, _ptr(ts.waypoints), _ptr(ts.offsets_dev), ts.offsets.ctypes.data_as(C.c_void_p),
                                                    2, _ptr(of_car), 0.9, 1.2, 0.17145 + 0.15875, 20.0, _ptr(st), 2 * len(poses), _ptr(out),
                                                    _ptr(ts.workspace), 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    got = out.cpu().numpy()
    assert np.array_equal(got[:len(poses)], _hip_plan(wp, poses, 0.9, 1.2)) and np.array_equal(got[len(poses):], act)


def test_hip_pure_pursuit_many_tracks_one_launch(assets):
    """f110_pure_pursuit_tracks: K racelines of different lengths (one beyond the LDS kernel's reach), every car on its
    own, in ONE launch `==` K single-raceline calls of f110_pure_pursuit; a second call re-uses the block boxes."""
    import torch
    from red_gym_amd import F110VecEnv, workload
    rng = np.random.default_rng(11)
    wp = _example_raceline(assets)
    lines = [wp, wp[::-1].copy(), wp[100:400].copy()]
    th = np.linspace(0, 2 * np.pi, 9000, endpoint=False)
    lines.append(np.stack([40 * np.cos(th) * (1 + 0.1 * np.sin(5 * th)), 25 * np.sin(th), rng.uniform(2, 8, 9000)], axis=1))
    B = 192
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False)
    assign = rng.integers(0, len(lines), B)          # any mixture: no blocks of envs needed
    poses = np.zeros((B, 1, 3))
    for b in range(B):
        ln = lines[assign[b]]
        k = rng.integers(0, len(ln))
        poses[b, 0] = [ln[k, 0] + rng.normal(0, 0.3), ln[k, 1] + rng.normal(0, 0.3), rng.uniform(-3, 3)]
    poses[:8, 0, :2] += 5.0
    env.reset(poses)
    ts, of_car = env.raceline_slots(lines, assign)
    act = env.pure_pursuit_tracks(ts, of_car, 0.9, 1.2).cpu().numpy()
    assert ts.boxes_valid
    act2 = env.pure_pursuit_tracks(ts, of_car, 0.9, 1.2).cpu().numpy()   # boxes re-used
    assert np.array_equal(act, act2)
    st = env.state.cpu().numpy()[:, 0]
    for k, ln in enumerate(lines):
        m = assign == k
        ref = _hip_plan(ln, st[m][:, [0, 1, 4]], 0.9, 1.2)
        assert np.array_equal(act[m, 0], ref), k
    # the cached convenience form used by examples/random_tracks.py
    act3 = env.pure_pursuit_blocks(lines, assign, 0.9, 1.2).cpu().numpy()
    assert np.array_equal(act3, act)
    env.close()


@pytest.mark.parametrize('M', [2, 3, 64, 65, 66, 130, 783, 4097, 5000, 6400, 6401, 8192, 20000])
def test_hip_pure_pursuit_raceline_lengths(M):
    """The planner kernel works through the raceline in 64-segment blocks (one wavefront per car, blocks skipped by
    their bounding box, a second mask word beyond 64 blocks): racelines of 2 ... 6400 points -- block boundaries, the
    >64-block path, the >64 KiB LDS path -- and beyond what LDS holds (6401 ... 20000 points: the global-memory form
    inside f110_pure_pursuit, every block evaluated) against the NumPy checker on poses on, near, off and far from
    the line; for the long ones also f110_pure_pursuit_tracks with its block boxes (`==` the former)."""
    import torch
    from argparse import Namespace
    from oracle.planner import PurePursuitPlanner, Raceline
    from red_gym_amd.engine import _lib, _ptr
    import ctypes as C
    rng = np.random.default_rng(M)
    th = np.linspace(0, 2 * np.pi, M, endpoint=False) if M > 3 else np.linspace(0, 1.0, M)
    R = 5.0 + 0.004 * M
    xy = np.stack([R * np.cos(th) * (1 + 0.2 * np.sin(3 * th)), 0.7 * R * np.sin(th)], axis=1)
    v = rng.uniform(2, 8, M)
    pl = PurePursuitPlanner.__new__(PurePursuitPlanner)
    pl.wheelbase, pl.max_reacquire, pl.line, pl.speeds = 0.33, 20., Raceline(xy), v
    n = 96
    k = rng.integers(0, M, n)
    poses = np.zeros((n, 7))
    poses[:, 0] = xy[k, 0] + rng.normal(0, 0.3, n)
    poses[:, 1] = xy[k, 1] + rng.normal(0, 0.3, n)
    poses[:16, :2] += rng.normal(0, 3.0, (16, 2))       # beyond the lookahead: re-acquire branch
    poses[16:20, :2] += 500.0                           # beyond max_reacquire
    poses[20:24, :2] = xy[[0, -1, M // 2, max(M - 2, 0)]]   # exactly on waypoints (first, last: the wrap search)
    poses[:, 4] = rng.uniform(-3, 3, n)
    lookahead = 0.9
    dev = torch.device('cuda', 0)
    wp = torch.as_tensor(np.column_stack([xy, v]), device=dev).contiguous()
    st = torch.as_tensor(poses, device=dev)
    out = torch.empty((n, 2), dtype=torch.float64, device=dev)
    lib = _lib.load()
    _lib.check(lib.f110_pure_pursuit(None, _ptr(wp), M, lookahead, 1.2, 0.33, 20.0, _ptr(st), n, _ptr(out),
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    got = out.cpu().numpy()
    for i in range(n):
        sp, stg = pl.plan(poses[i, 0], poses[i, 1], poses[i, 4], lookahead, 1.2)
        assert abs(got[i, 0] - stg) < 1e-12 and abs(got[i, 1] - sp) < 1e-12, (M, i, got[i], (stg, sp))
    with pytest.raises(ValueError):
        _lib.check(lib.f110_pure_pursuit(None, _ptr(wp), 1, lookahead, 1.2, 0.33, 20.0, _ptr(st), n, _ptr(out), None))
    if M >= 4097:
        from red_gym_amd.engine import TrackSet
        ts = TrackSet([np.column_stack([xy, v])], dev)
        out2 = torch.empty((n, 2), dtype=torch.float64, device=dev)
        _lib.check(lib.f110_pure_pursuit_tracks(None, _ptr(ts.waypoints), _ptr(ts.offsets_dev), ts.offsets.ctypes.data_as(C.c_void_p), 1,
                                                None, lookahead, 1.2, 0.33, 20.0, _ptr(st), n, _ptr(out2), _ptr(ts.workspace), 0,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        assert np.array_equal(out2.cpu().numpy(), got)
