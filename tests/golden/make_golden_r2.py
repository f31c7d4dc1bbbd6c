"""Round-2 golden fixtures, generated in the dev container by RUNNING THE REFERENCE's own code (never shipped;
/root/reference does not exist on the GPU box).  One fixture per fresh interpreter (the reference keeps its
scanner in class statics):

    python tests/golden/make_golden_r2.py            # all
    python tests/golden/make_golden_r2.py g9         # one

  g9_env2.npz        F110Env (gym/f110_gym/envs/f110_env.py) with TWO agents, both driven by the reference's
                     pure-pursuit planner (examples/waypoint_follow.py), until all(toggles >= 4): every step's
                     actions, poses, toggles, lap counts / times, done, collisions (f110_env.py:202-244 with A > 1:
                     the ego's start rotation applied to every car, frozen lap times, done through all()).
  g10_bitmap_calls.npz  weap_util/weap_util/lidar.py run with a RECORDING stand-in for cv2 (cv2 is not installed):
                     the integer arguments the reference hands to fillPoly / polylines / line / rectangle for 48
                     scans x the option grid.  Pins everything up to the OpenCV calls (beam subset :63-64, angle
                     table :67, points :70-73, centre rectangle :90-91); the rasterisation itself stays unpinned.
  g11_centerline.npz create_track() of gym/f110_gym/unittest/random_trackgen.py up to the shapely call (:56-159)
                     for 10 seeds x 6 consecutive calls: the centre line handed to shp.Polygon, or "gave up".
  g12_pointgrid.npz  main() of f1tenth_gym/examples/lidar.py (the routine that wrote the reference's
                     lidar_datasets/*.npz) run headless for 5 episodes on the older package copy it sits next to:
                     the scans it saw and the dataset it handed to np.savez_compressed (:212-254).
"""
import os
import runpy
import subprocess
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = '/root/reference'


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print('wrote', name, '%.1f KB' % (os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------------------------- g9
def g9_env2():
    import ref_loader
    import make_golden as mg  # noqa: F401  (its module body loads the reference env through ref_loader)
    bc, fe = mg.bc, mg.fe
    mod, conf = mg.load_planner()
    planner = mod.PurePursuitPlanner(conf, 0.17145 + 0.15875)
    tlad = 0.82461887897713965
    vgains = (1.375, 1.3)
    rl = mg.raceline()
    bc.RaceCar.scan_simulator = None
    env = fe.F110Env(map=conf.map_path, map_ext=conf.map_ext, num_agents=2, timestep=0.01,
                     integrator=bc.Integrator.RK4)
    # ego at the example's start pose; the second car 150 raceline rows (~30 m) ahead, where the track points
    # another way: its lap logic runs on the EGO's start rotation (f110_env.py:219-221)
    start = np.array([[conf.sx, conf.sy, conf.stheta], [rl[150, 1], rl[150, 2], rl[150, 3] + np.pi / 2]])
    obs, r, done, info = env.reset(start)
    keys = ('actions', 'x', 'y', 'theta', 'vx', 'col', 'lap_t', 'lap_c', 'toggle', 'near', 'done', 'ckpt', 'time')
    rec = {k: [] for k in keys}
    reset_obs = np.array([obs['poses_x'], obs['poses_y'], obs['poses_theta']]).T.copy()
    scans, scan_steps = [], []
    t0, k = time.time(), 0
    while not np.all(env.toggle_list >= 4) and k < 6000:
        act = np.zeros((2, 2))
        for a in range(2):
            sp, stg = planner.plan(obs['poses_x'][a], obs['poses_y'][a], obs['poses_theta'][a], tlad, vgains[a])
            act[a] = [stg, sp]
        obs, r, done, info = env.step(act)
        rec['actions'].append(act)
        rec['x'].append(list(obs['poses_x'])); rec['y'].append(list(obs['poses_y'])); rec['theta'].append(list(obs['poses_theta']))
        rec['vx'].append(list(obs['linear_vels_x'])); rec['col'].append(obs['collisions'].copy())
        rec['lap_t'].append(obs['lap_times'].copy()); rec['lap_c'].append(obs['lap_counts'].copy())
        rec['toggle'].append(env.toggle_list.copy()); rec['near'].append(np.array(env.near_starts, dtype=bool).copy())
        rec['done'].append(bool(done)); rec['ckpt'].append(np.array(info['checkpoint_done']).copy())
        rec['time'].append(env.current_time)
        if k % 500 == 0:
            scans.append(np.stack(obs['scans']).copy()); scan_steps.append(k)
            print('  step', k, 'toggles', env.toggle_list, 'wall %.0fs' % (time.time() - t0), flush=True)
        k += 1
    print('  2-agent closed loop: steps', k, 'toggles', env.toggle_list, 'lap times', env.lap_times,
          'collisions', np.array(rec['col']).sum(axis=0), 'first done', int(np.argmax(rec['done'])))
    save('g9_env2.npz', start=start, reset_obs=reset_obs, start_rot=np.asarray(env.start_rot), vgains=np.array(vgains),
         scan_steps=np.array(scan_steps), scans=np.array(scans), **{k_: np.array(v) for k_, v in rec.items()})


# --------------------------------------------------------------------------------------------- g10
class _Captured(Exception):
    pass


def g10_bitmap_calls():
    calls = []
    cv2 = types.ModuleType('cv2')  # records what the reference asks OpenCV to draw; draws nothing

    def _ints(a):
        return np.asarray(a).astype(np.int64).reshape(-1)
    cv2.fillPoly = lambda img, pts, color: calls.append(('fillPoly', np.concatenate([_ints(p) for p in pts]), int(color)))
    cv2.polylines = lambda img, pts, isClosed, color, thickness: calls.append(
        ('polylines', np.concatenate([_ints(p) for p in pts]), int(color) | (int(bool(isClosed)) << 16) | (int(thickness) << 20)))
    cv2.line = lambda img, p0, p1, color, thickness: calls.append(('line', np.concatenate([_ints(p0), _ints(p1)]), int(color)))
    cv2.rectangle = lambda img, p0, p1, color, thickness: calls.append(
        ('rectangle', np.concatenate([_ints(p0), _ints(p1)]), int(color) | ((int(thickness) & 0xff) << 20)))
    sys.modules['cv2'] = cv2
    ns = runpy.run_path(REF + '/weap_util/weap_util/lidar.py')
    ref_fn = ns['lidar_to_bitmap']
    # scans: reference scans of the example map (from g1/g8, themselves reference outputs) + synthetic shapes
    g1 = np.load(os.path.join(HERE, 'g1_scan.npz'))
    g8 = np.load(os.path.join(HERE, 'g8_env.npz'))
    rng = np.random.default_rng(77)
    scans = [g1['ex_scans'][i] for i in range(0, 40, 2)] + [g8['scans'][i] for i in range(0, len(g8['scans']), 3)][:12]
    th = np.linspace(0, 2 * np.pi, 1080)
    scans += [np.full(1080, 30.0), np.full(1080, 0.0), 12.8 + 0 * th, 5 + 4 * np.sin(3 * th), rng.uniform(0, 30, 1080),
              rng.uniform(0, 14, 1080), 12.75 + rng.uniform(-0.1, 0.1, 1080), np.abs(20 * np.cos(th)) + 0.5]
    scans += [g1['ex47_scans'][i] for i in range(8)]
    scans = np.array(scans[:48], dtype=np.float64)
    grid = [dict(draw_mode='FILL', bg_color='black'), dict(draw_mode='FILL', bg_color='white', draw_center=False),
            dict(draw_mode='POLYGON'), dict(draw_mode='POLYGON', winding_dir='CW', starting_angle=0.3, channels=3),
            dict(draw_mode='RAYS', target_beam_count=40, bg_color='black'),
            dict(draw_mode='FILL', max_scan_radius=30.0, scaling_factor=None, output_image_dims=(128, 200), channels=4),
            dict(draw_mode='POLYGON', fov=4.7, starting_angle=-np.pi / 2 - 4.7 / 2, target_beam_count=333, scaling_factor=25.5),
            dict(draw_mode='FILL', target_beam_count=1079, scaling_factor=4.25, output_image_dims=(255, 257))]
    ops, arg_off, arg_val, colors, scan_id, opt_id, shapes = [], [0], [], [], [], [], []
    code = {'fillPoly': 0, 'polylines': 1, 'line': 2, 'rectangle': 3}
    for oi, opt in enumerate(grid):
        for si, sc in enumerate(scans):
            del calls[:]
            img = ref_fn(list(sc), **opt)
            shapes.append(img.shape + (0,) * (3 - img.ndim))
            for name, ints, col in calls:
                ops.append(code[name]); arg_val.append(ints); arg_off.append(arg_off[-1] + len(ints))
                colors.append(col); scan_id.append(si); opt_id.append(oi)
    import json
    save('g10_bitmap_calls.npz', scans=scans, options=np.array(json.dumps(grid)), op=np.array(ops, np.int8),
         arg_off=np.array(arg_off, np.int64), arg_val=np.concatenate(arg_val).astype(np.int32),
         color=np.array(colors, np.int32), scan_id=np.array(scan_id, np.int16), opt_id=np.array(opt_id, np.int8),
         out_shape=np.array(shapes, np.int32))
    print('  %d draw calls recorded for %d scans x %d option sets' % (len(ops), len(scans), len(grid)))


# --------------------------------------------------------------------------------------------- g11
def g11_centerline():
    import tempfile
    seeds = [123, 7, 0, 1, 2, 3, 42, 99, 2025, 31337]
    CALLS = 6
    out = {}
    for seed in seeds:
        sink = []
        cv2 = types.ModuleType('cv2')
        shapely = types.ModuleType('shapely')
        geom = types.ModuleType('shapely.geometry')

        def Polygon(xy, _sink=sink):  # the centre line leaves create_track here (random_trackgen.py:161)
            _sink.append(np.array(xy, dtype=np.float64))
            raise _Captured()
        geom.Polygon = Polygon
        shapely.geometry = geom
        sys.modules.update({'cv2': cv2, 'shapely': shapely, 'shapely.geometry': geom})
        import matplotlib
        matplotlib.use('Agg')
        cwd, argv = os.getcwd(), sys.argv
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)  # the module creates maps/ and centerline/ in the working directory
        sys.argv = ['random_trackgen.py', '--seed', str(seed), '--num_maps', '1']
        try:
            # run_name != '__main__': the module body (argparse, np.random.seed(seed)) runs, its main loop does not
            ns = runpy.run_path(REF + '/gym/f110_gym/unittest/random_trackgen.py', run_name='ref_trackgen')
            status = []
            for c in range(CALLS):
                n0 = len(sink)
                try:
                    res = ns['create_track']()
                    assert res is False
                    status.append(0)           # gave up (:137 or :158)
                except _Captured:
                    assert len(sink) == n0 + 1
                    status.append(1)
        finally:
            os.chdir(cwd)
            sys.argv = argv
        out['seed%d_status' % seed] = np.array(status, np.int8)
        for j, c in enumerate(sink):
            out['seed%d_line%d' % (seed, j)] = c
        print('  seed', seed, 'status', status, 'lengths', [len(c) for c in sink])
    save('g11_centerline.npz', seeds=np.array(seeds), **out)


# --------------------------------------------------------------------------------------------- g12
def g12_pointgrid():
    import importlib
    import ref_loader
    # the older package copy next to the example (fov 4.7 default, f1tenth_gym/gym/f110_gym/envs/base_classes.py:68)
    ref_loader.REF_PKG = REF + '/f1tenth_gym/gym/f110_gym'
    lm, dm, cm, bc, fe = ref_loader.load_env()
    gym = sys.modules['gym']
    seen, saved = [], []

    def make(_id, **kw):
        bc.RaceCar.scan_simulator = None
        env = fe.F110Env(**kw)
        env.add_render_callback = lambda cb: None
        env.render = lambda mode='human': None
        step0, reset0, in_reset = env.step, env.reset, [False]

        def step(action):
            r = step0(action)
            if not in_reset[0]:  # F110Env.reset steps once itself (f110_env.py:335-336): not a dataset sample
                seen.append(r[0]['scans'][0].copy())
            return r

        def reset(poses):
            in_reset[0] = True
            try:
                return reset0(poses)
            finally:
                in_reset[0] = False
        env.step, env.reset = step, reset
        return env
    gym.make = make
    real_savez, real_sleep = np.savez_compressed, time.sleep

    def savez(filename, **kw):
        saved.append(kw['data'].copy())
        raise _Captured()
    cwd = os.getcwd()
    os.chdir(REF + '/f1tenth_gym/examples')  # relative config / map paths of the example
    real_makedirs = os.makedirs
    os.makedirs = lambda *a, **k: None      # the example creates lidar_datasets/ next to itself: not here
    np.savez_compressed, time.sleep = savez, (lambda s: None)
    np.random.seed(20250126)
    try:
        ns = runpy.run_path(REF + '/f1tenth_gym/examples/lidar.py', run_name='ref_lidar_example')
        try:
            ns['main']()
        except _Captured:
            pass
    finally:
        np.savez_compressed, time.sleep, os.makedirs = real_savez, real_sleep, real_makedirs
        os.chdir(cwd)
    data = saved[0]
    scans = np.array(seen)
    assert data.shape[0] == scans.shape[0] and data.dtype == np.uint8
    print('  %d samples, occupied cells per sample %.1f' % (len(data), data.reshape(len(data), -1).sum(1).mean()))
    save('g12_pointgrid.npz', scans=scans, data_bits=np.packbits(data, axis=-1), shape=np.array(data.shape))


ALL = {'g9': g9_env2, 'g10': g10_bitmap_calls, 'g11': g11_centerline, 'g12': g12_pointgrid}

if __name__ == '__main__':
    names = sys.argv[1:]
    if len(names) == 1 and names[0] in ALL:
        os.chdir(REF + '/examples')
        t = time.time()
        ALL[names[0]]()
        print('   %.1fs' % (time.time() - t))
    else:
        for nm in (names or list(ALL)):
            subprocess.run([sys.executable, os.path.abspath(__file__), nm], check=True)
