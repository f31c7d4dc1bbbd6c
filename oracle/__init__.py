"""CPU oracle for the F110Env.step hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end of oracle/f110_oracle.c (a scalar fp64 restatement of the
reference's gym/f110_gym/envs/{laser,dynamic,collision}_models.py,
base_classes.py and f110_env.py lap logic).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this package.

Pinned by tests/test_oracle_golden.py against vectors generated from the
reference itself (tests/golden/make_golden.py) and the reference's own KATs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'f110_oracle.c')
_BUILD = os.path.join(_HERE, '_build')
_SO = os.path.join(_BUILD, 'libf110_oracle.so')

PARAM_KEYS = ['mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 's_min', 's_max', 'sv_min',
              'sv_max', 'v_switch', 'a_max', 'v_min', 'v_max', 'width', 'length']
# f110_env.py:128
DEFAULT_PARAMS = {'mu': 1.0489, 'C_Sf': 4.718, 'C_Sr': 5.4562, 'lf': 0.15875, 'lr': 0.17145,
                  'h': 0.074, 'm': 3.74, 'I': 0.04712, 's_min': -0.4189, 's_max': 0.4189,
                  'sv_min': -3.2, 'sv_max': 3.2, 'v_switch': 7.319, 'a_max': 9.51, 'v_min': -5.0,
                  'v_max': 20.0, 'width': 0.31, 'length': 0.58}
RK4, EULER = 1, 2


def build(force=False):
    """gcc -O2, no FMA contraction, OpenMP for the batch baseline helper."""
    os.makedirs(_BUILD, exist_ok=True)
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    cmd = ['gcc', '-O2', '-std=c11', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math',
           '-fopenmp', '-o', _SO, _SRC, '-lm']
    subprocess.run(cmd, check=True)
    return _SO


class _Map(C.Structure):
    _fields_ = [('height', C.c_int), ('width', C.c_int), ('resolution', C.c_double),
                ('orig_x', C.c_double), ('orig_y', C.c_double), ('orig_c', C.c_double),
                ('orig_s', C.c_double), ('dt', C.c_void_p)]


class _ScanCfg(C.Structure):
    _fields_ = [('num_beams', C.c_int), ('theta_dis', C.c_int), ('fov', C.c_double),
                ('eps', C.c_double), ('max_range', C.c_double),
                ('theta_index_increment', C.c_double), ('sines', C.c_void_p),
                ('cosines', C.c_void_p)]


class _Car(C.Structure):
    _fields_ = [('state', C.c_double * 7), ('steer_buffer', C.c_double * 2),
                ('steer_count', C.c_int), ('in_collision', C.c_int), ('noise_step', C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_get_range.restype = C.c_double
        _lib.orc_env_sizeof.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def params_vec(params=None):
    params = DEFAULT_PARAMS if params is None else params
    return np.array([params[k] for k in PARAM_KEYS], dtype=np.float64)


def load_map(map_path, map_ext):
    """laser_models.py:383-427 set_map, restated (PIL + yaml + scipy EDT)."""
    import yaml
    from PIL import Image
    from scipy.ndimage import distance_transform_edt as edt
    img_path = os.path.splitext(map_path)[0] + map_ext
    img = np.array(Image.open(img_path).transpose(Image.FLIP_TOP_BOTTOM)).astype(np.float64)
    img[img <= 128.] = 0.
    img[img > 128.] = 255.
    with open(map_path, 'r') as f:
        meta = yaml.safe_load(f)
    res = meta['resolution']
    origin = meta['origin']
    return {'height': img.shape[0], 'width': img.shape[1], 'resolution': float(res),
            'orig_x': float(origin[0]), 'orig_y': float(origin[1]),
            'orig_s': float(np.sin(origin[2])), 'orig_c': float(np.cos(origin[2])),
            'dt': np.ascontiguousarray(res * edt(img)), 'img': img}


class Scanner(object):
    """ScanSimulator2D (laser_models.py:348-457) + RaceCar beam tables
    (base_classes.py:123-156)."""

    def __init__(self, num_beams=1080, fov=2 * np.pi, eps=0.0001, theta_dis=2000, max_range=30.0,
                 params=None):
        self.num_beams, self.fov, self.eps = num_beams, float(fov), eps
        self.theta_dis, self.max_range = theta_dis, max_range
        self.angle_increment = self.fov / (num_beams - 1)
        self.theta_index_increment = theta_dis * self.angle_increment / (2. * np.pi)
        theta_arr = np.linspace(0.0, 2 * np.pi, num=theta_dis)
        self.sines = np.ascontiguousarray(np.sin(theta_arr))
        self.cosines = np.ascontiguousarray(np.cos(theta_arr))
        self.cfg = _ScanCfg(num_beams, theta_dis, self.fov, eps, max_range,
                            self.theta_index_increment, _p(self.sines), _p(self.cosines))
        self.map = None
        self.cmap = None
        p = DEFAULT_PARAMS if params is None else params
        self.scan_angles = np.zeros(num_beams)
        self.beam_cosines = np.zeros(num_beams)
        self.side_distances = np.zeros(num_beams)
        lib().orc_beam_tables(C.c_int(num_beams), C.c_double(self.fov), C.c_double(p['width']),
                              C.c_double(p['lf']), C.c_double(p['lr']), _p(self.scan_angles),
                              _p(self.beam_cosines), _p(self.side_distances))

    def set_map(self, map_path, map_ext):
        self.set_map_dict(load_map(map_path, map_ext))
        return True

    def set_map_dict(self, m):
        self.map = m
        self.cmap = _Map(m['height'], m['width'], m['resolution'], m['orig_x'], m['orig_y'],
                         m['orig_c'], m['orig_s'], _p(m['dt']))

    def scan_batch(self, poses, return_lookups=False):
        if self.cmap is None:
            raise ValueError('Map is not set for scan simulator.')
        poses = _f64(poses).reshape(-1, 3)
        n = poses.shape[0]
        out = np.empty((n, self.num_beams))
        lk = np.zeros(n, dtype=np.int64)
        lib().orc_scan_batch(_p(poses), C.c_int(n), C.byref(self.cfg), C.byref(self.cmap), _p(out), _p(lk))
        return (out, lk) if return_lookups else out

    def beam_indices(self, pose):
        pose = _f64(pose)
        scan = np.empty(self.num_beams)
        idx = np.empty(self.num_beams, dtype=np.int32)
        lk = C.c_int64(0)
        lib().orc_get_scan(_p(pose), C.byref(self.cfg), C.byref(self.cmap), _p(scan), _p(idx), C.byref(lk))
        return idx

    def check_ttc(self, scan, vel, thresh=0.005):
        scan = _f64(scan)
        return bool(lib().orc_check_ttc(_p(scan), C.c_double(vel), _p(self.beam_cosines),
                                        _p(self.side_distances), C.c_double(thresh),
                                        C.c_int(self.num_beams)))

    def ray_cast(self, pose, scan, vertices):
        pose, scan, vertices = _f64(pose), _f64(scan).copy(), _f64(vertices)
        lib().orc_ray_cast(_p(pose), _p(scan), _p(self.scan_angles), C.c_int(self.num_beams), _p(vertices))
        return scan

    def blocked_view_indices(self, pose, vertices):
        pose, vertices = _f64(pose), _f64(vertices)
        lo, hi = C.c_int(0), C.c_int(0)
        lib().orc_get_blocked_view_indices(_p(pose), _p(vertices), _p(self.scan_angles),
                                           C.c_int(self.num_beams), C.byref(lo), C.byref(hi))
        return lo.value, hi.value


def vehicle_dynamics_st(x, u, pvec):
    x, u, f = _f64(x), _f64(u), np.empty(7)
    lib().orc_vehicle_dynamics_st(_p(x), _p(u), _p(pvec), _p(f))
    return f


def vehicle_dynamics_ks(x, u, pvec):
    x, u, f = _f64(x), _f64(u), np.empty(5)
    lib().orc_vehicle_dynamics_ks(_p(x), _p(u), _p(pvec), _p(f))
    return f


def pid(speed, steer, current_speed, current_steer, max_sv, max_a, max_v, min_v):
    a, s = C.c_double(0), C.c_double(0)
    lib().orc_pid(*[C.c_double(v) for v in (speed, steer, current_speed, current_steer, max_sv,
                                             max_a, max_v, min_v)], C.byref(a), C.byref(s))
    return a.value, s.value


def update_pose_batch(states, steer_buffers, steer_counts, actions, pvec, time_step, integrator):
    """n independent RaceCar.update_pose calls (without the scan).
    Returns (new_states[n,7], new_steer_buffers[n,2], new_counts[n])."""
    states = _f64(states)
    n = states.shape[0]
    cars = (_Car * n)()
    for i in range(n):
        cars[i].state[:] = states[i]
        cars[i].steer_buffer[:] = steer_buffers[i]
        cars[i].steer_count = int(steer_counts[i])
    actions = _f64(actions)
    lib().orc_update_pose_batch(cars, C.c_int(n), _p(actions), _p(pvec), C.c_double(time_step),
                                C.c_int(integrator))
    ns = np.array([list(c.state) for c in cars])
    nb = np.array([list(c.steer_buffer) for c in cars])
    nc = np.array([c.steer_count for c in cars])
    return ns, nb, nc


def get_vertices(pose, length, width):
    pose, out = _f64(pose), np.empty((4, 2))
    lib().orc_get_vertices(_p(pose), C.c_double(length), C.c_double(width), _p(out))
    return out


def collision(v1, v2):
    v1, v2 = _f64(v1), _f64(v2)
    return bool(lib().orc_collision_quads(_p(v1), _p(v2)))


def collision_multiple(vertices):
    vertices = _f64(vertices)
    n = vertices.shape[0]
    col, idx = np.empty(n), np.empty(n)
    lib().orc_collision_multiple(_p(vertices), C.c_int(n), _p(col), _p(idx))
    return col, idx


def get_range(pose, beam_theta, va, vb):
    pose, va, vb = _f64(pose), _f64(va), _f64(vb)
    return lib().orc_get_range(_p(pose), C.c_double(beam_theta), _p(va), _p(vb))


def noise_table(seed, steps, num_beams=1080, std_dev=0.01):
    """Block k = the k-th `rng.normal(0., std_dev, size=num_beams)` draw of
    default_rng(seed) (base_classes.py:202, laser_models.py:451)."""
    rng = np.random.default_rng(seed=seed)
    return np.ascontiguousarray(np.stack([rng.normal(0., std_dev, size=num_beams) for _ in range(steps)]))


class Env(object):
    """F110Env + Simulator for ONE env (f110_env.py:261-347, base_classes.py:546-623)."""

    def __init__(self, scanner, num_agents=1, params=None, time_step=0.01, integrator=RK4,
                 ego_idx=0, noise=None, _buf=None):
        self.scanner = scanner
        self.num_agents = num_agents
        self.pvec = params_vec(params)
        self.noise = None if noise is None else _f64(noise)
        self._buf = _buf if _buf is not None else np.zeros(lib().orc_env_sizeof(), dtype=np.uint8)
        self._ptr = _p(self._buf)
        lib().orc_env_init(self._ptr, C.c_int(num_agents), C.c_int(ego_idx), C.c_int(integrator),
                           C.c_double(time_step), _p(self.pvec), C.byref(scanner.cfg),
                           C.byref(scanner.cmap), _p(scanner.scan_angles), _p(scanner.beam_cosines),
                           _p(scanner.side_distances),
                           None if self.noise is None else _p(self.noise),
                           C.c_int64(0 if self.noise is None else self.noise.shape[0]))
        self.scans = np.zeros((num_agents, scanner.num_beams))

    def reset(self, poses):
        poses = _f64(poses)
        done = lib().orc_env_reset(self._ptr, _p(poses), _p(self.scans))
        return self.observe(bool(done))

    def step(self, action):
        action = _f64(action)
        done = lib().orc_env_step(self._ptr, _p(action), _p(self.scans))
        return self.observe(bool(done))

    def sim_step(self, action):
        """Simulator.step only (no lap logic)."""
        action = _f64(action)
        lib().orc_sim_step(self._ptr, _p(action), _p(self.scans))
        return self.observe(False)

    def set_state(self, agent, state, steer_buffer=(0., 0.), steer_count=2, noise_step=0):
        state, sb = _f64(state), _f64(steer_buffer)
        lib().orc_env_set_state(self._ptr, C.c_int(agent), _p(state), _p(sb), C.c_int(steer_count),
                                C.c_int64(noise_step))

    def observe(self, done):
        A = self.num_agents
        st = np.empty((A, 7))
        col, idx, lt, lc, tg = (np.empty(A) for _ in range(5))
        ct, lk = C.c_double(0), C.c_int64(0)
        lib().orc_env_get(self._ptr, _p(st), _p(col), _p(idx), _p(lt), _p(lc), _p(tg), C.byref(ct), C.byref(lk))
        return {'scans': self.scans.copy(), 'state': st, 'collisions': col, 'collision_idx': idx,
                'lap_times': lt, 'lap_counts': lc, 'toggles': tg, 'current_time': ct.value,
                'lookups': lk.value, 'done': done}


class Batch(object):
    """B independent Envs with next-step auto-reset (cpu_baseline leg)."""

    def __init__(self, scanner, num_envs, num_agents, spawn, params=None, time_step=0.01,
                 integrator=RK4, noise=None):
        self.B, self.A = num_envs, num_agents
        sz = lib().orc_env_sizeof()
        self._buf = np.zeros(num_envs * sz, dtype=np.uint8)
        self.noise = None if noise is None else _f64(noise)
        self.envs = [Env(scanner, num_agents, params, time_step, integrator, 0, self.noise,
                         _buf=self._buf[i * sz:(i + 1) * sz]) for i in range(num_envs)]
        self.spawn = _f64(spawn).reshape(num_envs, num_agents, 3)
        self.pending = np.ones(num_envs, dtype=np.uint8)
        self.scans = np.zeros((num_envs, num_agents, scanner.num_beams))

    def step(self, actions, threads=1):
        actions = _f64(actions)
        lib().orc_batch_step(_p(self._buf), C.c_int(self.B), _p(actions), _p(self.spawn),
                             _p(self.pending), _p(self.scans), C.c_int(threads))
        return self.pending.copy()
