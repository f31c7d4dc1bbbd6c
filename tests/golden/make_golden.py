"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE
(/root/reference, imported through ref_loader.py) in the dev container.

    python tests/golden/make_golden.py [names...]

Outputs are data only (inputs + expected outputs, np.savez_compressed, fp64).
The reference itself never travels; tests load the .npz files.

Fixtures:
  g1_scan.npz      ScanSimulator2D.scan (noise off) + per-beam LUT indices + lookup counts
  g2_noise.npz     default_rng(12345).normal(0, 0.01, 1080) blocks
  g3_dynamics.npz  RaceCar.update_pose (scan stubbed) RK4 + Euler, pid, vehicle_dynamics_st/ks
  g4_gjk.npz       get_vertices, collision, collision_multiple
  g5_ttc.npz       check_ttc_jit + beam tables
  g6_raycast.npz   ray_cast, get_blocked_view_indices
  g7_sim.npz       Simulator.step trajectories (1 and 2 agents)
  g8_env.npz       F110Env 2-lap closed loop with the reference's pure-pursuit caller
"""
import importlib.util
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

REF = ref_loader.REF_ROOT
EX_MAP = REF + '/examples/example_map'
MAPS = REF + '/gym/f110_gym/envs/maps/'
lm, dm, cm, bc, fe = ref_loader.load_env()

PARAMS = {'mu': 1.0489, 'C_Sf': 4.718, 'C_Sr': 5.4562, 'lf': 0.15875, 'lr': 0.17145, 'h': 0.074,
          'm': 3.74, 'I': 0.04712, 's_min': -0.4189, 's_max': 0.4189, 'sv_min': -3.2, 'sv_max': 3.2,
          'v_switch': 7.319, 'a_max': 9.51, 'v_min': -5.0, 'v_max': 20.0, 'width': 0.31, 'length': 0.58}
PKEYS = ['mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 's_min', 's_max', 'sv_min', 'sv_max',
         'v_switch', 'a_max', 'v_min', 'v_max']


def raceline():
    return np.loadtxt(REF + '/examples/example_waypoints.csv', delimiter=';', skiprows=3)


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print('wrote', name, '%.1f KB' % (os.path.getsize(path) / 1024))


class Instrument(object):
    """Records int(theta_index) per beam and counts distance-table reads by
    wrapping the reference's own module-level functions."""

    def __enter__(self):
        self.idx, self.lookups = [], 0
        self._tr, self._dt = lm.trace_ray, lm.distance_transform

        def tr(x, y, theta_index, *a):
            self.idx.append(int(theta_index))
            return self._tr(x, y, theta_index, *a)

        def dtf(*a):
            self.lookups += 1
            return self._dt(*a)
        lm.trace_ray, lm.distance_transform = tr, dtf
        return self

    def __exit__(self, *a):
        lm.trace_ray, lm.distance_transform = self._tr, self._dt


def scan_set(sim, poses):
    scans, idxs, lks = [], [], []
    for p in poses:
        with Instrument() as ins:
            scans.append(sim.scan(np.array(p), None))
        idxs.append(np.array(ins.idx, dtype=np.int32))
        lks.append(ins.lookups)
    return np.array(scans), np.array(idxs), np.array(lks, dtype=np.int64)


def g1_scan():
    rl = raceline()
    rng = np.random.default_rng(101)
    out = {}
    # example_map, fov 2pi: raceline poses (heading psi + pi/2) with jitter
    rows = rng.choice(rl.shape[0], size=40, replace=False)
    poses = np.stack([rl[rows, 1], rl[rows, 2], rl[rows, 3] + np.pi / 2], axis=1)
    poses[8:] += np.stack([rng.normal(0, 0.2, 32), rng.normal(0, 0.2, 32),
                           np.clip(rng.normal(0, 0.2, 32), -0.5, 0.5)], axis=1)
    # special yaws (exact LUT-index boundaries) + poses leaving / outside the map + inside a wall
    special = np.array([[0.7, 0.0, np.pi], [0.7, 0.0, 0.0], [0.7, 0.0, 2 * np.pi], [0.7, 0.0, -0.3],
                        [0.7, 0.0, 7.0], [0.7, 0.0, np.pi / 2], [0.7, 0.0, 1.37079632679],
                        [20.0, 50.0, 0.3], [-78.0, -44.0, 1.0], [21.7, 55.5, 4.0], [-90.0, 0.0, 0.0],
                        [0.7, 200.0, 2.0], [-30.0, -44.3, 2.5], [21.78, 0.0, 0.1]])
    sim = lm.ScanSimulator2D(1080, 2 * np.pi)
    sim.set_map(EX_MAP + '.yaml', '.png')
    allp = np.concatenate([poses, special])
    s, i, l = scan_set(sim, allp)
    out.update(ex_poses=allp, ex_scans=s, ex_idx=i, ex_lookups=l)
    # example_map, fov 4.7 (older copy's default)
    sim47 = lm.ScanSimulator2D(1080, 4.7)
    sim47.set_map(EX_MAP + '.yaml', '.png')
    s, i, l = scan_set(sim47, allp[:12])
    out.update(ex47_poses=allp[:12], ex47_scans=s, ex47_idx=i, ex47_lookups=l)
    # a different beam count / theta_dis / max_range / eps
    simx = lm.ScanSimulator2D(271, 4.7, eps=0.001, theta_dis=1500, max_range=12.0)
    simx.set_map(EX_MAP + '.yaml', '.png')
    s, i, l = scan_set(simx, allp[:6])
    out.update(exx_poses=allp[:6], exx_scans=s, exx_idx=i, exx_lookups=l)
    # berlin / skirk / vegas at the legacy poses (unittest/scan_sim.py:321-366, legacy_scan_gen.py)
    legacy = np.stack([np.zeros(10), np.zeros(10), np.linspace(-1., 1., num=10)], axis=1)
    for name in ('berlin', 'skirk', 'vegas'):
        sm = lm.ScanSimulator2D(1080, 4.7)
        sm.set_map(MAPS + name + '.yaml', '.png')
        n = 10 if name != 'vegas' else 4
        s, i, l = scan_set(sm, legacy[:n])
        out.update({name + '_poses': legacy[:n], name + '_scans': s, name + '_idx': i, name + '_lookups': l})
        out[name + '_dt_corner'] = sm.dt[-1, -1]
    out['ex_dt_corner'] = sim.dt[-1, -1]
    # the distance table itself is pinned through a checksum + sampled cells
    rr = rng.integers(0, sim.dt.shape[0], 4096)
    cc = rng.integers(0, sim.dt.shape[1], 4096)
    out.update(ex_dt_rows=rr, ex_dt_cols=cc, ex_dt_vals=sim.dt[rr, cc], ex_dt_sum=np.sum(sim.dt))
    save('g1_scan.npz', **out)


def g2_noise():
    out = {}
    for seed in (12345, 0, 7):
        rng = np.random.default_rng(seed=seed)
        out['seed%d' % seed] = np.stack([rng.normal(0., 0.01, size=1080) for _ in range(8)])
    save('g2_noise.npz', **out)


def make_car(integrator, state, buf):
    car = bc.RaceCar(PARAMS, 12345, time_step=0.01, integrator=integrator)
    car.reset(np.zeros(3))  # only the first RaceCar ever built gets scan_rng in __init__ (base_classes.py:116-117)
    car.state = np.array(state, dtype=np.float64)
    car.steer_buffer = np.array(buf, dtype=np.float64)
    return car


def g3_dynamics():
    rng = np.random.default_rng(303)
    # stub the scanner: update_pose's dynamics part only (base_classes.py:254-402)
    if bc.RaceCar.scan_simulator is None:
        bc.RaceCar(PARAMS, 12345)

    class _NoScan(object):
        def scan(self, pose, rng_):
            return None
    real = bc.RaceCar.scan_simulator
    bc.RaceCar.scan_simulator = _NoScan()
    try:
        n = 3000
        st = np.zeros((n, 7))
        st[:, 0] = rng.uniform(-50, 50, n)
        st[:, 1] = rng.uniform(-50, 50, n)
        st[:, 2] = rng.uniform(-0.45, 0.45, n)
        st[:, 3] = rng.uniform(-6, 21, n)
        st[:, 4] = rng.uniform(-0.2, 2 * np.pi + 0.2, n)
        st[:, 5] = rng.uniform(-3, 3, n)
        st[:, 6] = rng.uniform(-0.5, 0.5, n)
        # enrich: low speed (kinematic branch + its switch), saturation, yaw wrap, steer at limits
        st[:600, 3] = rng.uniform(-0.7, 0.7, 600)
        st[600:700, 3] = rng.choice([0.0, 0.5, -0.5, 0.49999999, 20.0, -5.0, 7.319, 7.4], 100)
        st[700:800, 2] = rng.choice([-0.4189, 0.4189, 0.0, 0.42, -0.42], 100)
        st[800:900, 4] = rng.choice([0.0, 2 * np.pi, 2 * np.pi - 1e-4, 1e-5, -1e-5], 100)
        act = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-6, 22, n)], axis=1)
        act[900:1000, 0] = st[900:1000, 2] + rng.choice([0., 1e-4, -1e-4, 5e-5, 2e-4], 100)
        cnt = rng.integers(0, 3, n)
        buf = rng.uniform(-0.4, 0.4, (n, 2))
        res = {}
        for integ, tag in ((bc.Integrator.RK4, 'rk4'), (bc.Integrator.Euler, 'euler')):
            ns, nb, nc = np.zeros((n, 7)), np.zeros((n, 2)), np.zeros(n, dtype=np.int64)
            for i in range(n):
                # reference buffer layout: np.append(raw, buf) -> newest first
                car = make_car(integ, st[i], buf[i, :cnt[i]])
                car.update_pose(act[i, 0], act[i, 1])
                ns[i] = car.state
                nc[i] = car.steer_buffer.shape[0]
                nb[i, :nc[i]] = car.steer_buffer
            res[tag + '_state'] = ns
            res[tag + '_buf'] = nb
            res[tag + '_cnt'] = nc
        # sequences from reset (steer FIFO): 40 cars x 12 steps
        seq_act = np.stack([rng.uniform(-0.4, 0.4, (40, 12)), rng.uniform(0, 8, (40, 12))], axis=2)
        seq_pose = np.stack([rng.uniform(-5, 5, 40), rng.uniform(-5, 5, 40), rng.uniform(0, 6.2, 40)], axis=1)
        seq_state = np.zeros((40, 12, 7))
        for c in range(40):
            car = bc.RaceCar(PARAMS, 12345, time_step=0.01, integrator=bc.Integrator.RK4)
            car.reset(seq_pose[c])
            for k in range(12):
                car.update_pose(seq_act[c, k, 0], seq_act[c, k, 1])
                seq_state[c, k] = car.state
        # raw RHS + pid
        pv = [PARAMS[k] for k in PKEYS]
        f_st = np.array([dm.vehicle_dynamics_st(st[i], act[i], *pv) for i in range(1000)])
        f_ks = np.array([dm.vehicle_dynamics_ks(st[i, :5], act[i], *pv) for i in range(1000)])
        pid_out = np.array([dm.pid(act[i, 1], act[i, 0], st[i, 3], st[i, 2], PARAMS['sv_max'],
                                   PARAMS['a_max'], PARAMS['v_max'], PARAMS['v_min']) for i in range(1000)])
        save('g3_dynamics.npz', state=st, action=act, cnt=cnt, buf=buf, seq_act=seq_act, seq_pose=seq_pose,
             seq_state=seq_state, f_st=f_st, f_ks=f_ks, pid=pid_out, **res)
    finally:
        bc.RaceCar.scan_simulator = real


def g4_gjk():
    rng = np.random.default_rng(404)
    L, W = PARAMS['length'], PARAMS['width']
    n = 6000
    pa = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(-7, 7, n)], axis=1)
    # partner within ~1 car length so that about half of the pairs overlap
    d = rng.uniform(0, 0.9, n)
    ang = rng.uniform(0, 2 * np.pi, n)
    pb = np.stack([pa[:, 0] + d * np.cos(ang), pa[:, 1] + d * np.sin(ang), rng.uniform(-7, 7, n)], axis=1)
    pb[:50] = pa[:50]                      # identical poses (zero centroid difference)
    pb[50:100, 2] = pa[50:100, 2]          # parallel, near-touching side by side
    pb[50:100, 0] = pa[50:100, 0] - (W + rng.uniform(-1e-3, 1e-3, 50)) * np.sin(pa[50:100, 2])
    pb[50:100, 1] = pa[50:100, 1] + (W + rng.uniform(-1e-3, 1e-3, 50)) * np.cos(pa[50:100, 2])
    va = np.array([cm.get_vertices(p, L, W) for p in pa])
    vb = np.array([cm.get_vertices(p, L, W) for p in pb])
    hit = np.array([cm.collision(np.ascontiguousarray(va[i]), np.ascontiguousarray(vb[i])) for i in range(n)])
    out = dict(pose_a=pa, pose_b=pb, verts_a=va, verts_b=vb, hit=hit)
    for A in (2, 3, 4):
        m = 300
        poses = np.zeros((m, A, 3))
        poses[:, :, 0] = rng.uniform(-0.6, 0.6, (m, A))
        poses[:, :, 1] = rng.uniform(-0.6, 0.6, (m, A))
        poses[:, :, 2] = rng.uniform(0, 6.3, (m, A))
        col, idx = np.zeros((m, A)), np.zeros((m, A))
        for k in range(m):
            allv = np.stack([cm.get_vertices(poses[k, a], L, W) for a in range(A)])
            col[k], idx[k] = cm.collision_multiple(allv)
        out.update({'multi%d_poses' % A: poses, 'multi%d_col' % A: col, 'multi%d_idx' % A: idx})
    # the reference's own KAT inputs (collision_models.py:306-324), legacy RandomState(1234)
    np.random.seed(1234)
    v1 = np.asarray([[4, 11.], [5, 5], [9, 9], [10, 10]])
    kat = [v1 + np.random.normal(size=v1.shape) / 100. for _ in range(6)] + [v1 + 10.]
    kc, ki = cm.collision_multiple(np.stack(kat))
    out.update(kat_verts=np.stack(kat), kat_col=kc, kat_idx=ki)
    save('g4_gjk.npz', **out)


def g5_ttc():
    rng = np.random.default_rng(505)
    if bc.RaceCar.scan_simulator is None or bc.RaceCar.scan_angles is None:
        bc.RaceCar(PARAMS, 12345)
    ang, cosv, side = bc.RaceCar.scan_angles, bc.RaceCar.cosines, bc.RaceCar.side_distances
    n = 400
    base = rng.uniform(0.2, 30, (n, 1080)).astype(np.float32)  # stored as f32 (exactly representable)
    scans = base.astype(np.float64)
    vel = rng.uniform(-5, 20, n)
    vel[:40] = 0.0
    # put one beam right at the 0.005 s threshold for most of them (override list, fp64)
    ov_beam = np.full(n, -1, dtype=np.int64)
    ov_val = np.zeros(n)
    for i in range(40, 340):
        b = rng.integers(0, 1080)
        pv = vel[i] * cosv[b]
        ov_beam[i] = b
        ov_val[i] = side[b] + pv * 0.005 * rng.choice([0.999999, 1.0, 1.000001, 0.5, 1.5, -0.1])
        scans[i, b] = ov_val[i]
    hit = np.array([lm.check_ttc_jit(scans[i], vel[i], ang, cosv, side, 0.005) for i in range(n)])
    save('g5_ttc.npz', scans_f32=base, ov_beam=ov_beam, ov_val=ov_val, vel=vel, hit=hit, scan_angles=ang,
         cosines=cosv, side_distances=side)


def g6_raycast():
    rng = np.random.default_rng(606)
    if bc.RaceCar.scan_angles is None:
        bc.RaceCar(PARAMS, 12345)
    ang = bc.RaceCar.scan_angles
    L, W = PARAMS['length'], PARAMS['width']
    n = 240
    ego = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(0, 2 * np.pi, n)], axis=1)
    dist = rng.uniform(0.5, 6, n)
    bearing = rng.uniform(-np.pi, np.pi, n)
    bearing[:60] = np.pi + rng.uniform(-0.05, 0.05, 60)  # opponent straddling +-pi behind the ego
    ego[60:80, 2] = 0.0                                   # post-iTTC zeroed yaw
    opp = np.stack([ego[:, 0] + dist * np.cos(ego[:, 2] + bearing), ego[:, 1] + dist * np.sin(ego[:, 2] + bearing),
                    rng.uniform(0, 2 * np.pi, n)], axis=1)
    scans_in32 = rng.uniform(0.3, 8, (n, 1080)).astype(np.float32)  # stored as f32 (exactly representable)
    scans_in = scans_in32.astype(np.float64)
    scans_out = np.zeros_like(scans_in)
    span = np.zeros((n, 2), dtype=np.int64)
    verts = np.zeros((n, 4, 2))
    for i in range(n):
        verts[i] = cm.get_vertices(opp[i], L, W)
        span[i] = lm.get_blocked_view_indices(ego[i], verts[i], ang)
        scans_out[i] = lm.ray_cast(ego[i], scans_in[i].copy(), ang, verts[i])
    save('g6_raycast.npz', ego=ego, opp=opp, verts=verts, scans_in_f32=scans_in32, scans_out=scans_out, span=span,
         scan_angles=ang)


def load_planner():
    spec = importlib.util.spec_from_file_location('ref_waypoint_follow', REF + '/examples/waypoint_follow.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from argparse import Namespace
    import yaml
    with open(REF + '/examples/config_example_map.yaml') as f:
        conf = Namespace(**yaml.safe_load(f))
    conf.wpt_path = REF + '/examples/example_waypoints.csv'
    conf.map_path = EX_MAP
    return mod, conf


def g7_sim():
    mod, conf = load_planner()
    planner = mod.PurePursuitPlanner(conf, 0.17145 + 0.15875)
    tlad, vgain = 0.82461887897713965, 1.375
    out = {}
    # --- 1 agent, 300 steps, pure pursuit then a hard left into the wall (iTTC hit, state zeroing)
    bc.RaceCar.scan_simulator = None
    sim = bc.Simulator(PARAMS, 1, 12345, 2 * np.pi, time_step=0.01, integrator=bc.Integrator.RK4)
    sim.set_map(EX_MAP + '.yaml', '.png')
    sim.reset(np.array([[conf.sx, conf.sy, conf.stheta]]))
    T = 300
    acts, states, cols, scans = np.zeros((T, 1, 2)), np.zeros((T, 1, 7)), np.zeros((T, 1)), []
    obs = sim.step(np.zeros((1, 2)))
    for k in range(T):
        sp, stg = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], tlad, vgain)
        if k >= 180:
            sp, stg = 6.0, 0.4
        acts[k, 0] = [stg, sp]
        obs = sim.step(acts[k])
        states[k, 0] = sim.agents[0].state
        cols[k] = obs['collisions']
        if k % 10 == 0 or cols[k].any():
            scans.append((k, obs['scans'][0].copy()))
    out.update(a1_start=np.array([[conf.sx, conf.sy, conf.stheta]]), a1_actions=acts, a1_states=states,
               a1_collisions=cols, a1_scan_steps=np.array([s[0] for s in scans]),
               a1_scans=np.array([s[1] for s in scans]))
    print('  1-agent: first wall hit at step', int(np.argmax(cols[:, 0] > 0)) if cols.any() else None)
    # --- 2 agents, 260 steps: ego rear-ends a slower opponent (GJK), both ray-cast each other
    bc.RaceCar.scan_simulator = None
    sim = bc.Simulator(PARAMS, 2, 12345, 2 * np.pi, time_step=0.01, integrator=bc.Integrator.RK4)
    sim.set_map(EX_MAP + '.yaml', '.png')
    rl = raceline()
    start = np.array([[conf.sx, conf.sy, conf.stheta],
                      [rl[12, 1], rl[12, 2], rl[12, 3] + np.pi / 2]])
    sim.reset(start)
    T = 260
    acts, states = np.zeros((T, 2, 2)), np.zeros((T, 2, 7))
    cols, cidx, scans = np.zeros((T, 2)), np.zeros((T, 2)), []
    obs = sim.step(np.zeros((2, 2)))
    for k in range(T):
        for a, vg in ((0, 1.375), (1, 0.45)):
            sp, stg = planner.plan(obs['poses_x'][a], obs['poses_y'][a], obs['poses_theta'][a], tlad, vg)
            acts[k, a] = [stg, sp]
        if k >= 220:
            acts[k, 1] = [-0.4, 5.0]
        obs = sim.step(acts[k])
        for a in range(2):
            states[k, a] = sim.agents[a].state
        cols[k] = obs['collisions']
        cidx[k] = sim.collision_idx
        if k % 5 == 0 or cols[k].any():
            scans.append((k, np.stack(obs['scans']).copy()))
    out.update(a2_start=start, a2_actions=acts, a2_states=states, a2_collisions=cols, a2_collision_idx=cidx,
               a2_scan_steps=np.array([s[0] for s in scans]), a2_scans=np.array([s[1] for s in scans]))
    print('  2-agent: collision steps', np.nonzero(cols.any(axis=1))[0][:10], 'idx', np.unique(cidx))
    save('g7_sim.npz', **out)


def g8_env():
    mod, conf = load_planner()
    planner = mod.PurePursuitPlanner(conf, 0.17145 + 0.15875)
    tlad, vgain = 0.82461887897713965, 1.375
    bc.RaceCar.scan_simulator = None
    env = fe.F110Env(map=conf.map_path, map_ext=conf.map_ext, num_agents=1, timestep=0.01,
                     integrator=bc.Integrator.RK4)
    start = np.array([[conf.sx, conf.sy, conf.stheta]])
    obs, r, done, info = env.reset(start)
    rec = {k: [] for k in ('actions', 'x', 'y', 'theta', 'vx', 'wz', 'col', 'lap_t', 'lap_c', 'toggle', 'done', 'state')}
    scans, scan_steps = [obs['scans'][0].copy()], [-1]
    reset_obs = np.array([obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0],
                          obs['lap_times'][0], obs['lap_counts'][0]])
    t0 = time.time()
    k = 0
    while not done:
        sp, stg = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], tlad, vgain)
        obs, r, done, info = env.step(np.array([[stg, sp]]))
        rec['actions'].append([stg, sp])
        rec['x'].append(obs['poses_x'][0]); rec['y'].append(obs['poses_y'][0]); rec['theta'].append(obs['poses_theta'][0])
        rec['vx'].append(obs['linear_vels_x'][0]); rec['wz'].append(obs['ang_vels_z'][0])
        rec['col'].append(obs['collisions'][0]); rec['lap_t'].append(obs['lap_times'][0])
        rec['lap_c'].append(obs['lap_counts'][0]); rec['toggle'].append(env.toggle_list[0]); rec['done'].append(done)
        rec['state'].append(env.sim.agents[0].state.copy())
        if k % 100 == 0 or done:
            scans.append(obs['scans'][0].copy()); scan_steps.append(k)
        k += 1
    print('  closed loop: steps', k, 'sim time', env.current_time, 'laps', env.lap_counts, 'wall', time.time() - t0)
    save('g8_env.npz', start=start, reset_obs=reset_obs, scan_steps=np.array(scan_steps), scans=np.array(scans),
         final_time=env.current_time, **{k_: np.array(v) for k_, v in rec.items()})


ALL = {'g1': g1_scan, 'g2': g2_noise, 'g3': g3_dynamics, 'g4': g4_gjk, 'g5': g5_ttc, 'g6': g6_raycast,
       'g7': g7_sim, 'g8': g8_env}

if __name__ == '__main__':
    os.chdir(REF + '/examples')
    names = sys.argv[1:] or list(ALL)
    for nm in names:
        t = time.time()
        print('==', nm)
        ALL[nm]()
        print('   %.1fs' % (time.time() - t))
