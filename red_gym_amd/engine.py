"""Engine: owns one f110_handle (one GPU), the torch tensors bound to it and the
host-built tables.  Thin: all compute is in libf110_hip.so."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .base_classes import integrator_code
from .maps import load_map

# f110_env.py:128
DEFAULT_PARAMS = {'mu': 1.0489, 'C_Sf': 4.718, 'C_Sr': 5.4562, 'lf': 0.15875, 'lr': 0.17145, 'h': 0.074,
                  'm': 3.74, 'I': 0.04712, 's_min': -0.4189, 's_max': 0.4189, 'sv_min': -3.2, 'sv_max': 3.2,
                  'v_switch': 7.319, 'a_max': 9.51, 'v_min': -5.0, 'v_max': 20.0, 'width': 0.31, 'length': 0.58}

_TORCH_DTYPES = {
    'state': torch.float64, 'steer_buf': torch.float64, 'steer_cnt': torch.int32, 'noise_step': torch.int32,
    'spawn': torch.float64, 'start_rot': torch.float64, 'near_start': torch.uint8, 'toggles': torch.int32,
    'current_time': torch.float64, 'pending_reset': torch.uint8, 'scans': torch.float32,
    'scans_f64': torch.float64, 'pose_snap': torch.float64, 'collisions': torch.uint8,
    'collision_idx': torch.int32, 'in_collision': torch.uint8, 'lap_counts': torch.int32,
    'lap_times': torch.float64, 'done': torch.bool, 'checkpoint_done': torch.bool, 'lookups': torch.int32,
}


def params_vec(params):
    return np.array([float(params[k]) for k in _lib.PARAM_KEYS], dtype=np.float64)


def beam_tables(num_beams, fov, params):
    """RaceCar.__init__ tables (base_classes.py:116-156) as whole-array NumPy: scan angles, their cosines and,
    per beam, the distance from the lidar to the car's outline (half width sideways, half wheelbase fore/aft)
    along that beam.  The reference picks the two candidate distances by quadrant; folded to the first
    quadrant (|angle|, then minus pi/2 beyond it) those are the same operands, so the values are identical
    bit for bit (tests/test_oracle_golden.py::test_beam_tables_match_reference)."""
    half_w = params['width'] / 2.
    half_l = (params['lf'] + params['lr']) / 2.
    scan_angles = -fov / 2. + np.arange(num_beams) * (fov / (num_beams - 1))
    cosines = np.cos(scan_angles)
    # the reference's branches: angle > 0 ? (angle < pi/2 ? angle : angle - pi/2) : (angle > -pi/2 ? -angle : -angle - pi/2)
    mag = np.where(scan_angles > 0, scan_angles, -scan_angles)
    rear = np.where(scan_angles > 0, scan_angles >= np.pi / 2, scan_angles <= -np.pi / 2)  # second / third quadrant
    folded = np.where(rear, mag - np.pi / 2., mag)
    # a beam at exactly 0 rad divides by sin(0) like the reference does (inf loses the minimum); only the
    # RuntimeWarning is silenced
    with np.errstate(divide='ignore'):
        s, c = np.sin(folded), np.cos(folded)
        to_side = half_w / np.where(rear, c, s)
        to_fr = half_l / np.where(rear, s, c)
    return scan_angles, cosines, np.minimum(to_side, to_fr)


class NoiseTable(object):
    """Rows of `default_rng(seed).normal(0, std, num_beams)` (laser_models.py:451,
    base_classes.py:202): every car draws one row per scan since its reset.  The host-side source of the noise
    (Engine(noise_source='numpy')): NumPy itself draws, the rows are uploaded.  The default source produces the same
    rows on the device (csrc/f110_noise.h)."""

    def __init__(self, seed, num_beams, std_dev=0.01):
        self.seed, self.num_beams, self.std_dev = seed, num_beams, std_dev
        self._rng = np.random.default_rng(seed=seed)
        self.rows = np.zeros((0, num_beams))

    def ensure(self, steps):
        if steps > self.rows.shape[0]:
            new = [self._rng.normal(0., self.std_dev, size=self.num_beams)
                   for _ in range(steps - self.rows.shape[0])]
            self.rows = np.ascontiguousarray(np.concatenate([self.rows, np.stack(new)], axis=0))
        return self.rows


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def on_own_device(method):
    """The library launches on the caller's stream and refuses a handle whose device is not the calling thread's
    current one (f110_hip.h, conventions): every launching method runs with the engine's device current, so that
    engines on different GPUs -- or next to the caller's own work on another GPU -- can share a process."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with torch.cuda.device(self.device):
            return method(self, *args, **kwargs)
    return wrapper


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine(object):
    NOISE_CHUNK = 256  # rows produced ahead of the cars at a time

    def __init__(self, num_envs=1, num_agents=1, params=None, seed=12345, fov=2 * np.pi, timestep=0.01,
                 integrator=1, ego_idx=0, num_beams=1080, eps=0.0001, theta_dis=2000, max_range=30.0,
                 ttc_thresh=0.005, device=0, autoreset=False, noise_std=0.01, noise_steps=0,
                 keep_f64_scans=False, count_lookups=False, noise_source='device'):
        """`params`: one dict (every env, f110_env.py:125-128) or a sequence of num_envs dicts (env e constructed with
        params[e]); `seed`: one int (:102-105) or a sequence of num_envs ints.
        `noise_source`: 'device' (rows produced on the GPU, kept once per distinct seed in a table that runs ahead of the
        cars: at most F110_MAX_NOISE_SLOTS distinct seeds), 'per_env' (every env its own generator, its row produced by the
        step itself: any number of seeds -- chosen by itself when 'device' is asked for more distinct seeds than the table
        holds) or 'numpy' (NumPy draws on the host, rows uploaded);
        `noise_steps`: rows to have ready at construction (0: the first step asks for them)."""
        if not torch.cuda.is_available():
            raise RuntimeError('red_gym_amd needs a HIP device (torch.cuda.is_available() is False); '
                               'there is no CPU path.')
        if noise_source == 'per_car':
            noise_source = 'per_env'   # (the cars of an env share its seed and therefore its rows)
        if noise_source not in ('device', 'numpy', 'per_env'):
            raise ValueError("noise_source must be 'device', 'per_env' or 'numpy'")
        self.lib = _lib.load()
        env_params = None
        if params is not None and not isinstance(params, dict):
            env_params = [dict(p) for p in params]
            if len(env_params) != int(num_envs):
                raise ValueError('params must be one dict or a sequence of num_envs (%d) dicts' % int(num_envs))
            params = env_params[0]
        env_seeds = None
        if not np.isscalar(seed) and seed is not None:
            env_seeds = [int(x) for x in seed]
            if len(env_seeds) != int(num_envs):
                raise ValueError('seed must be one int or a sequence of num_envs (%d) ints' % int(num_envs))
            seed = env_seeds[0]
        self.params = dict(DEFAULT_PARAMS if params is None else params)
        self.autoreset = bool(autoreset)
        self.B, self.A, self.num_beams = int(num_envs), int(num_agents), int(num_beams)
        self.N = self.B * self.A
        self.fov, self.timestep, self.seed = float(fov), float(timestep), seed
        self.device_index = int(device)
        self.device = torch.device('cuda', self.device_index)
        cfg = _lib.Config()
        cfg.num_envs, cfg.num_agents, cfg.num_beams, cfg.theta_dis = self.B, self.A, self.num_beams, int(theta_dis)
        cfg.integrator, cfg.ego_idx = integrator_code(integrator), int(ego_idx)
        cfg.device, cfg.autoreset = self.device_index, int(bool(autoreset))
        cfg.fov, cfg.eps, cfg.max_range = self.fov, float(eps), float(max_range)
        cfg.timestep, cfg.ttc_thresh = self.timestep, float(ttc_thresh)
        cfg.params[:] = list(params_vec(self.params))
        self._h = C.c_void_p()
        _lib.check(self.lib.f110_create(C.byref(cfg), C.byref(self._h)))
        # tables computed with numpy exactly as the reference does (laser_models.py:379-381)
        theta_arr = np.linspace(0.0, 2 * np.pi, num=int(theta_dis))
        self.sines = np.ascontiguousarray(np.sin(theta_arr))
        self.cosines = np.ascontiguousarray(np.cos(theta_arr))
        self.scan_angles, self.beam_cosines, self.side_distances = beam_tables(self.num_beams, self.fov, self.params)
        _lib.check(self.lib.f110_set_tables(self._h, _np_ptr(self.sines), _np_ptr(self.cosines),
                                            _np_ptr(self.scan_angles), _np_ptr(self.beam_cosines),
                                            _np_ptr(self.side_distances)))
        self.map = None
        self.slot_shapes = {}
        self.host_steps_bound = 0
        self._steps_exact = False
        self._noise_on = bool(noise_std and noise_std > 0)
        self._noise_gen = self._noise_on and noise_source == 'device'
        self._noise_per_env = self._noise_on and noise_source == 'per_env'
        self._noise_rows, self._noise_floor, self._noise_prefetched = 0, 0, False
        self._in_capture = False  # a stream capture is recording step(): no noise work (it was done in front of the capture)
        self.scan_reorder = True  # keep the scan's launch order sorted by noise row (Engine._reorder_scan); False: car order
        self.env_noise_assign = None
        self.noise_tables = []
        if env_params is not None:
            self.set_env_params(env_params)
        if self._noise_on:
            self._setup_noise(env_seeds if env_seeds is not None else [seed], float(noise_std))
        self._alloc(keep_f64_scans, count_lookups)
        if self._noise_on and noise_steps:
            with torch.cuda.device(self.device):
                self._noise_to(int(noise_steps))

    # ------------------------------------------------------------------ buffers
    def _alloc(self, keep_f64, count_lookups):
        B, A, N, nb, dev = self.B, self.A, self.N, self.num_beams, self.device
        shapes = {'state': (B, A, 7), 'steer_buf': (B, A, 2), 'steer_cnt': (B, A), 'noise_step': (B, A),
                  'spawn': (B, A, 3), 'start_rot': (B, 4), 'near_start': (B, A), 'toggles': (B, A),
                  'current_time': (B,), 'pending_reset': (B,), 'scans': (B, A, nb), 'scans_f64': (B, A, nb),
                  'pose_snap': (B, A, 3), 'collisions': (B, A), 'collision_idx': (B, A), 'in_collision': (B, A),
                  'lap_counts': (B, A), 'lap_times': (B, A), 'done': (B,), 'checkpoint_done': (B, A), 'lookups': (B, A)}
        self.t = {}
        bufs = _lib.Buffers()
        for name in _lib.BUFFER_FIELDS:
            if (name == 'scans_f64' and not keep_f64) or (name == 'lookups' and not count_lookups):
                self.t[name] = None
                setattr(bufs, name, None)
                continue
            self.t[name] = torch.zeros(shapes[name], dtype=_TORCH_DTYPES[name], device=dev)
            setattr(bufs, name, self.t[name].data_ptr())
        self.t['start_rot'][:, 0] = 1.
        self.t['start_rot'][:, 3] = 1.
        self.t['near_start'].fill_(1)
        self.t['collision_idx'].fill_(-1)
        torch.cuda.synchronize(dev)
        _lib.check(self.lib.f110_bind(self._h, C.byref(bufs)))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ per-env constructor arguments
    def set_env_params(self, env_params):
        """env e drives the vehicle of env_params[e] (a sequence of num_envs dicts): what num_envs reference envs
        constructed with their own `params` would do.  Equal dicts share a params slot."""
        if len(env_params) != self.B:
            raise ValueError('one params dict per env (%d), got %d' % (self.B, len(env_params)))
        vecs = np.stack([params_vec(p) for p in env_params])
        uniq, inv = np.unique(vecs, axis=0, return_inverse=True)
        # slot 0 must be env 0's (the handle's constructor params: beam tables, defaults)
        first = int(np.asarray(inv).reshape(-1)[0])
        order = [first] + [k for k in range(uniq.shape[0]) if k != first]
        remap = np.empty(uniq.shape[0], dtype=np.int32)
        remap[order] = np.arange(uniq.shape[0], dtype=np.int32)
        table = np.ascontiguousarray(uniq[order], dtype=np.float64)
        assign = np.ascontiguousarray(remap[np.asarray(inv).reshape(-1)], dtype=np.int32)
        _lib.check(self.lib.f110_set_params_slots(self._h, _np_ptr(table), table.shape[0]))
        _lib.check(self.lib.f110_assign_params(self._h, _np_ptr(assign) if table.shape[0] > 1 else None))
        self.env_params_assign = assign

    def _setup_noise(self, seeds, std):
        """One noise slot per distinct seed (base_classes.py:117,202: all cars of an env draw from default_rng(seed))."""
        uniq = list(dict.fromkeys(seeds))
        if len(uniq) > _lib.F110_MAX_NOISE_SLOTS and self._noise_gen:
            self._noise_gen, self._noise_per_env = False, True   # more seeds than the table has slots: every env its own generator
        if self._noise_per_env:
            # f110_env.py:102-105: every env its own `seed`; np.random.PCG64(seed).state is the stream default_rng(seed) starts from
            m64 = (1 << 64) - 1
            words = np.empty((self.B, 4), dtype=np.uint64)
            per_seed = {}
            for e in range(self.B):
                sd = seeds[e if len(seeds) > 1 else 0]
                if sd not in per_seed:
                    st = np.random.PCG64(sd).state['state']
                    per_seed[sd] = (st['state'] & m64, st['state'] >> 64, st['inc'] & m64, st['inc'] >> 64)
                words[e] = per_seed[sd]
            _lib.check(self.lib.f110_set_noise_per_env(self._h, _np_ptr(words), std))
            self.noise_seeds, self._noise_std = uniq, std
            return
        if len(uniq) > _lib.F110_MAX_NOISE_SLOTS:
            raise ValueError("%d distinct seeds; a table of host rows holds %d noise slots (noise_source='per_env' has no limit)"
                             % (len(uniq), _lib.F110_MAX_NOISE_SLOTS))
        self.noise_seeds, self._noise_std = uniq, std
        for k, sd in enumerate(uniq):
            if self._noise_gen:
                st = np.random.PCG64(sd).state['state']  # the stream np.random.default_rng(sd) starts from
                m64 = (1 << 64) - 1
                words = (C.c_uint64 * 4)(st['state'] & m64, st['state'] >> 64, st['inc'] & m64, st['inc'] >> 64)
                _lib.check(self.lib.f110_set_noise_generator(self._h, k, words, std))
            else:
                self.noise_tables.append(NoiseTable(sd, self.num_beams, std))
        if not self._noise_gen:
            self._upload_host_noise(64)
        if len(seeds) > 1:
            assign = np.ascontiguousarray([uniq.index(sd) for sd in seeds], dtype=np.int32)
            _lib.check(self.lib.f110_assign_noise(self._h, _np_ptr(assign) if len(uniq) > 1 else None))
            if len(uniq) > 1:
                self.env_noise_assign = torch.as_tensor(assign, device=self.device)

    def _upload_host_noise(self, rows):
        for k, nt in enumerate(self.noise_tables):
            tbl = nt.ensure(rows)
            _lib.check(self.lib.f110_set_noise_slot(self._h, k, _np_ptr(tbl), tbl.shape[0]))
        self._noise_rows = rows

    def noise_info(self):
        """(floor, rows produced, ring rows per slot, slots, bytes held) of the device noise table."""
        lo, hi, cap, nbytes = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        slots = C.c_int32(0)
        _lib.check(self.lib.f110_noise_info(self._h, C.byref(lo), C.byref(hi), C.byref(cap), C.byref(slots), C.byref(nbytes)))
        return lo.value, hi.value, cap.value, slots.value, nbytes.value

    def noise_rows(self, slot, row0, n_rows):
        """Rows of the device noise table as a NumPy array [n_rows, num_beams] (tests / diagnostics; synchronises)."""
        out = np.empty((int(n_rows), self.num_beams), dtype=np.float64)
        _lib.check(self.lib.f110_noise_read(self._h, int(slot), int(row0), int(n_rows), _np_ptr(out)))
        return out

    def _noise_to(self, rows):
        """Rows 0 .. rows-1 (above the floor) readable by the steps enqueued from now on."""
        if self._noise_per_env:
            return
        if self._noise_gen:
            _lib.check(self.lib.f110_noise_ensure(self._h, int(rows), self._stream()))
            self._noise_rows = self.noise_info()[1]  # (host-side bookkeeping of the library: no synchronisation)
        elif rows > self._noise_rows:
            self._upload_host_noise(max(int(rows), 2 * self._noise_rows))
        self._noise_prefetched = False

    def _set_noise_floor(self, lo):
        if self._noise_gen and lo != self._noise_floor:
            _lib.check(self.lib.f110_noise_set_floor(self._h, int(lo), self._stream()))
            if lo < self._noise_floor:
                # the dropped rows are produced again from the generators' marks (in this stream); what was produced stays
                self._noise_rows = self.noise_info()[1]
                self._noise_prefetched = False
            self._noise_floor = int(lo)

    @on_own_device
    def _ensure_noise(self, may_raise_floor=True):
        """Keeps the noise table ahead of every car without synchronising: host_steps_bound counts steps since the last
        full reset, an upper bound of every car's noise_step.  While the rows produced cover it, nothing happens (but the
        next chunk is started on the library's side stream once half of the current one is used).  When the bound reaches
        the table, the true (min, max) of the counters is fetched -- one small D2H copy per chunk of steps, none at all
        when autoreset is off and no env was reset on its own, since every car then stands exactly at the bound -- when
        the next chunk would not fit the table and cars cannot go back to row 0 by themselves (autoreset off), the floor
        moves up to the slowest car instead of the table growing (a ring of constant size however long the run), and the
        next chunk is produced.  may_raise_floor=False: a reset is about to send cars back to row 0."""
        if not self._noise_on or self._in_capture or self._noise_per_env:
            return   # (per-env noise: the step produces its own rows)
        self._reorder_scan()
        need = self.host_steps_bound + 2
        if need <= self._noise_rows:
            if self._noise_gen and not self._noise_prefetched and need + self.NOISE_CHUNK // 2 > self._noise_rows:
                _lib.check(self.lib.f110_noise_prefetch(self._h, self._noise_rows + self.NOISE_CHUNK))
                self._noise_prefetched = True
            return
        if self._steps_exact:
            mn = mx = self.host_steps_bound
        else:
            mn, mx = (int(v) for v in torch.aminmax(self.t['noise_step']))
            self.host_steps_bound = mx
        target = mx + 2 + self.NOISE_CHUNK
        if (self._noise_gen and may_raise_floor and not self.autoreset and mn > self._noise_floor
                and target - self._noise_floor > self.noise_info()[2]):
            self._set_noise_floor(mn)
        if mx + 2 > self._noise_rows and self._noise_prefetched:
            self._noise_to(mx + 2)   # the chunk that was produced beside the steps joins the table (a stream wait, no kernel)
        if mx + 2 > self._noise_rows:
            self._noise_to(target)   # not covered by a prefetch: produced now, in the caller's stream

    REORDER_EVERY = 64     # steps between two sorts of the scan's launch order
    REORDER_MIN_CARS = 65536  # (measured: +2.3 % in the steady state at 65 536 cars, neutral right after a reset; -1.3 % at 32 768 cars)

    def _reorder_scan(self):
        """Keeps the scan's launch order sorted by the envs' noise-row counters (f110_set_scan_order): envs that were reset at
        different times stand on different rows of the noise table, and a launch in car order then streams one 8.6 KB row per
        env from HBM; launched side by side, the envs of one row share it in the L1 / L2 (10 % of the step at 65 536 envs).  All
        counters advance together, so the order only ages through resets: one device-side sort every REORDER_EVERY steps
        (~50 us) keeps it.  Results do not depend on the order.  Nothing to do while every car stands on the same row."""
        if self._steps_exact or self.N < self.REORDER_MIN_CARS or not self.scan_reorder:
            return
        self._reorder_count = getattr(self, '_reorder_count', 0) + 1
        if self._reorder_count % self.REORDER_EVERY != 1:
            return
        key = self.t['noise_step'][:, 0]
        if getattr(self, 'env_noise_assign', None) is not None:
            key = key.to(torch.int64) + (self.env_noise_assign.to(torch.int64) << 32)   # (envs of one seed AND one row together)
        idx = torch.argsort(key)
        if self.A > 1:
            idx = (idx.unsqueeze(1) * self.A + torch.arange(self.A, device=self.device)).reshape(-1)
        # (Measured and dropped: giving every XCD one contiguous EIGHTH of the sorted list, so that its L2 holds an eighth of the
        # rows -- the steady state does not move (103.7 M) and the protocol region loses 4 % (105.2 against 109.4 M): age
        # correlates with what a car's scan costs, and an XCD that is dealt the expensive eighth finishes last.)
        if getattr(self, '_scan_order', None) is None:
            self._scan_order = idx.to(torch.int32).contiguous()
            _lib.check(self.lib.f110_set_scan_order(self._h, _ptr(self._scan_order)))
        else:
            self._scan_order.copy_(idx)

    def device_errors(self):
        """The handle's device error word (f110_device_errors; synchronises): 0 = none, bit 0 = a car's noise row was
        outside the table, bit 1 = an index check of the bounds-checked build failed."""
        flags = C.c_uint32(0)
        _lib.check(self.lib.f110_device_errors(self._h, C.byref(flags)))
        return flags.value

    # ------------------------------------------------------------------ map / params
    def set_map(self, map_path, map_ext):
        m = load_map(map_path, map_ext)
        self.set_map_data(m)
        return True

    def set_map_data(self, m):
        _lib.check(self.lib.f110_set_map_occupancy(self._h, _np_ptr(m.free), m.height, m.width, m.resolution,
                                                   m.orig_x, m.orig_y, m.orig_c, m.orig_s))
        self.map = m

    def set_map_occupancy(self, free, resolution, orig_x, orig_y, orig_theta=0.0, slot=0):
        """Map from an occupancy mask (nonzero = free, row 0 = bottom row, as after laser_models.py:399-404).
        A CUDA uint8 tensor stays on the device: EDT, cell codes, LUT and the fp64 table are built there (~1 ms),
        so a generated track can be installed every episode."""
        import torch
        oc, os_ = float(np.cos(orig_theta)), float(np.sin(orig_theta))
        if torch.is_tensor(free) and free.is_cuda:
            f = free.to(device=self.device, dtype=torch.uint8).contiguous()
            if not bool((f == 0).any()):
                raise ValueError('map has no occupied cell')
            H, W = f.shape
            _lib.check(self.lib.f110_set_map_slot_occupancy_dev(self._h, int(slot), f.data_ptr(), H, W, float(resolution),
                                                                float(orig_x), float(orig_y), oc, os_))
        else:
            f = np.ascontiguousarray(np.asarray(free) != 0, dtype=np.uint8)
            H, W = f.shape
            _lib.check(self.lib.f110_set_map_slot_occupancy(self._h, int(slot), _np_ptr(f), H, W, float(resolution),
                                                            float(orig_x), float(orig_y), oc, os_))
        self.slot_shapes[int(slot)] = (H, W)
        if slot == 0:
            self.map = ('dt', (H, W))

    def assign_maps(self, map_of_env=None):
        """Gives every env its map slot (int array [num_envs] of used slots 0.._lib.F110_MAX_MAPS-1; None: all on slot 0;
        IndexError for a slot that holds no map).  Any pattern works; the scan keeps its full occupancy when maps go to
        blocks of envs with an even car count (the two cars of a scan workgroup then share one copy of their map's table),
        and runs one wave per workgroup otherwise (a map per env: about a quarter slower, same bits)."""
        if map_of_env is None:
            _lib.check(self.lib.f110_assign_maps(self._h, None))
            return
        m = np.ascontiguousarray(map_of_env, dtype=np.int32)
        if m.shape != (self.B,):
            raise ValueError('map_of_env must have one entry per env (%d), got shape %s' % (self.B, m.shape))
        _lib.check(self.lib.f110_assign_maps(self._h, _np_ptr(m)))

    def set_map_dt(self, dt, resolution, orig_x, orig_y, orig_c=1.0, orig_s=0.0):
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        _lib.check(self.lib.f110_set_map_dt(self._h, _np_ptr(dt), dt.shape[0], dt.shape[1], float(resolution),
                                            float(orig_x), float(orig_y), float(orig_c), float(orig_s)))
        self.map = ('dt', dt.shape)

    def get_map_dt(self, slot=0):
        if slot == 0:
            if self.map is None:
                raise ValueError('Map is not set for scan simulator.')
            shape = (self.map.height, self.map.width) if not isinstance(self.map, tuple) else self.map[1]
        else:
            if slot not in self.slot_shapes:
                raise ValueError('Map is not set for scan simulator.')
            shape = self.slot_shapes[slot]
        out = np.empty(shape, dtype=np.float64)
        _lib.check(self.lib.f110_get_map_slot_dt(self._h, int(slot), _np_ptr(out)))
        return out

    def update_params(self, params, agent_idx=-1):
        """base_classes.py:507-527: all agents (agent_idx < 0) or one agent of every env."""
        pv = params_vec(params)
        _lib.check(self.lib.f110_update_params(self._h, _np_ptr(pv), int(agent_idx)))
        if agent_idx < 0:
            self.params = dict(params)

    # ------------------------------------------------------------------ step path
    @on_own_device
    def reset(self, poses, mask=None):
        """poses: [B,A,3] float64 tensor on the device; mask: [B] uint8 tensor or None."""
        if tuple(poses.shape) != (self.B, self.A, 3):
            raise ValueError('Number of poses for reset does not match number of agents.')
        poses = poses.to(device=self.device, dtype=torch.float64).contiguous()
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            self._steps_exact = False      # some cars go back to row 0, the others run on
        else:
            self.host_steps_bound = 0
            self._steps_exact = not self.autoreset
        if self._noise_floor > 0:
            self._set_noise_floor(0)       # cars at row 0 again: the rows below the floor are produced anew
        self._ensure_noise(may_raise_floor=False)
        _lib.check(self.lib.f110_reset(self._h, _ptr(poses), _ptr(mask), self._stream()))
        self.host_steps_bound += 1
        self._keep = (poses, mask)  # keep inputs alive until the stream has consumed them

    @on_own_device
    def step(self, actions):
        """actions: [B,A,2] float64 tensor on the device (steer, speed)."""
        if tuple(actions.shape) != (self.B, self.A, 2):
            raise ValueError('actions must have shape (%d, %d, 2)' % (self.B, self.A))
        if actions.dtype != torch.float64 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float64).contiguous()
        self._ensure_noise()
        _lib.check(self.lib.f110_step(self._h, _ptr(actions), self._stream()))
        self.host_steps_bound += 1
        self._keep = actions

    # ------------------------------------------------------------------ function-level entry points
    def _dev64(self, a, shape=None):
        t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)) if not torch.is_tensor(a) else a
        t = t.to(device=self.device, dtype=torch.float64).contiguous()
        return t.reshape(shape) if shape is not None else t

    @on_own_device
    def scan(self, poses, want_f32=False, want_lookups=False):
        poses = self._dev64(poses, (-1, 3))
        n = poses.shape[0]
        out = torch.empty((n, self.num_beams), dtype=torch.float64, device=self.device)
        out32 = torch.empty((n, self.num_beams), dtype=torch.float32, device=self.device) if want_f32 else None
        lk = torch.zeros((n,), dtype=torch.int32, device=self.device) if want_lookups else None
        _lib.check(self.lib.f110_scan(self._h, _ptr(poses), n, _ptr(out), _ptr(out32), _ptr(lk), self._stream()))
        res = [out]
        if want_f32:
            res.append(out32)
        if want_lookups:
            res.append(lk)
        return res[0] if len(res) == 1 else tuple(res)

    @on_own_device
    def update_pose(self, state, steer_buf, steer_cnt, actions):
        state, steer_buf = self._dev64(state, (-1, 7)).clone(), self._dev64(steer_buf, (-1, 2)).clone()
        cnt = torch.as_tensor(np.asarray(steer_cnt)).to(device=self.device, dtype=torch.int32).contiguous().clone()
        actions = self._dev64(actions, (-1, 2))
        _lib.check(self.lib.f110_update_pose(self._h, _ptr(state), _ptr(steer_buf), _ptr(cnt), _ptr(actions),
                                             state.shape[0], self._stream()))
        return state, steer_buf, cnt

    @on_own_device
    def vehicle_dynamics(self, x, u, kinematic=False):
        """dynamic_models.py:124-176 (or :91-121): f(x, u) with agent 0's parameters."""
        x, u = self._dev64(x, (-1, 7)), self._dev64(u, (-1, 2))
        f = torch.empty_like(x)
        _lib.check(self.lib.f110_vehicle_dynamics(self._h, _ptr(x), _ptr(u), x.shape[0], int(bool(kinematic)), _ptr(f),
                                                  self._stream()))
        return f

    @on_own_device
    def get_vertices(self, poses):
        poses = self._dev64(poses, (-1, 3))
        out = torch.empty((poses.shape[0], 4, 2), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.f110_get_vertices(self._h, _ptr(poses), poses.shape[0], _ptr(out), self._stream()))
        return out

    @on_own_device
    def gjk_pairs(self, va, vb):
        va, vb = self._dev64(va, (-1, 4, 2)), self._dev64(vb, (-1, 4, 2))
        hit = torch.zeros((va.shape[0],), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.f110_gjk_pairs(self._h, _ptr(va), _ptr(vb), va.shape[0], _ptr(hit), self._stream()))
        return hit

    @on_own_device
    def collision_multiple(self, verts):
        verts = self._dev64(verts)
        n, A = verts.shape[0], verts.shape[1]
        col = torch.zeros((n, A), dtype=torch.uint8, device=self.device)
        idx = torch.zeros((n, A), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.f110_collision_multiple(self._h, _ptr(verts), n, A, _ptr(col), _ptr(idx), self._stream()))
        return col, idx

    @on_own_device
    def check_ttc(self, scans, vel):
        scans, vel = self._dev64(scans, (-1, self.num_beams)), self._dev64(vel, (-1,))
        hit = torch.zeros((scans.shape[0],), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.f110_check_ttc(self._h, _ptr(scans), _ptr(vel), scans.shape[0], _ptr(hit), self._stream()))
        return hit

    @on_own_device
    def ray_cast(self, ego_poses, opp_verts, scans):
        ego, verts = self._dev64(ego_poses, (-1, 3)), self._dev64(opp_verts, (-1, 4, 2))
        scans = self._dev64(scans, (-1, self.num_beams)).clone()
        span = torch.zeros((ego.shape[0], 2), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.f110_ray_cast(self._h, _ptr(ego), _ptr(verts), ego.shape[0], _ptr(scans), _ptr(span),
                                          self._stream()))
        return scans, span

    # ------------------------------------------------------------------ planner
    @on_own_device
    def pure_pursuit(self, waypoints, lookahead, vgain, wheelbase=0.17145 + 0.15875, max_reacquire=20.,
                     state=None, out=None, prepare=True):
        """waypoints: [M,3] (x, y, speed) device tensor; returns actions [B,A,2] (steer, speed)
        planned from the current state -- feed it straight to step().  prepare: a raceline tensor that is planned on
        a second time (same storage, unchanged since: torch's version counter) is prepared once (f110_pure_pursuit_prepare:
        a grid of candidate segments, ~0.1 s on the host) and from then on planned with one lane per car."""
        st = self.t['state'] if state is None else self._dev64(state, (-1, 7))
        n = st.numel() // 7
        if out is None:
            out = torch.empty((n, 2), dtype=torch.float64, device=self.device)
        if not (torch.is_tensor(waypoints) and waypoints.device == self.device and waypoints.dtype == torch.float64
                and waypoints.is_contiguous()):
            waypoints = self._dev64(waypoints, (-1, 3))
        elif prepare and not self._in_capture and 2 <= waypoints.shape[0] <= 65535:
            key = (waypoints.data_ptr(), waypoints._version, waypoints.shape[0])
            if getattr(self, '_plan_seen', None) == key and getattr(self, '_plan_key', None) != key:
                _lib.check(self.lib.f110_pure_pursuit_prepare(self._h, _ptr(waypoints), waypoints.shape[0], 0.0, 0.0, self._stream()))
                self._plan_key = key
            elif getattr(self, '_plan_key', None) is not None and self._plan_key != key and self._plan_key[0] == key[0]:
                # same storage, new values: the prepared grid is stale -- forget it (prepared again on the next call)
                _lib.check(self.lib.f110_pure_pursuit_prepare(self._h, _ptr(waypoints), waypoints.shape[0], 0.0, 0.0, self._stream()))
                self._plan_key = key
            self._plan_seen = key
        _lib.check(self.lib.f110_pure_pursuit(self._h, _ptr(waypoints), waypoints.shape[0], float(lookahead),
                                              float(vgain), float(wheelbase), float(max_reacquire), _ptr(st), n,
                                              _ptr(out), self._stream()))
        self._keep_wp = waypoints
        return out.view(self.B, self.A, 2) if state is None else out

    @on_own_device
    def pure_pursuit_tracks(self, tracks, track_of_car, lookahead, vgain, wheelbase=0.17145 + 0.15875, max_reacquire=20.,
                            state=None, out=None):
        """Pure pursuit when cars drive on different racelines (one launch): `tracks` a TrackSet, `track_of_car` an
        int32 device tensor [n] (None: raceline 0 for all)."""
        st = self.t['state'] if state is None else self._dev64(state, (-1, 7))
        n = st.numel() // 7
        if out is None:
            out = torch.empty((n, 2), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.f110_pure_pursuit_tracks(self._h, _ptr(tracks.waypoints), _ptr(tracks.offsets_dev),
                                                     _np_ptr(tracks.offsets), tracks.K, _ptr(track_of_car), float(lookahead),
                                                     float(vgain), float(wheelbase), float(max_reacquire), _ptr(st), n, _ptr(out),
                                                     _ptr(tracks.workspace), int(tracks.boxes_valid), self._stream()))
        tracks.boxes_valid = True
        self._keep_tracks = (tracks, track_of_car)
        return out.view(self.B, self.A, 2) if state is None else out

    @on_own_device
    def pack_env(self, env=0):
        """Env `env`'s observation as ONE pinned host fp64 row (f110_pack_env + one device -> host copy + a stream
        synchronisation): [A*7 state | A collisions | A lap_times | A lap_counts | A toggles | current_time | done | A*nb scans].
        The returned NumPy array is a view of a buffer that the next call overwrites."""
        if getattr(self, '_pack_dev', None) is None:
            n = int(self.lib.f110_pack_env_size(self._h))
            self._pack_dev = torch.empty((n,), dtype=torch.float64, device=self.device)
            self._pack_host = torch.empty((n,), dtype=torch.float64).pin_memory()
        _lib.check(self.lib.f110_pack_env(self._h, int(env), _ptr(self._pack_dev), self._stream()))
        self._pack_host.copy_(self._pack_dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._pack_host.numpy()

    def set_scan_stages(self, spec=None):
        """Wave -> car mapping of the scan launches (f110_set_scan_stages): e.g. '*:-2,6144:0,2048:2'."""
        _lib.check(self.lib.f110_set_scan_stages(self._h, None if spec is None else spec.encode()))

    def launch_epoch(self):
        """Changes whenever a captured hipGraph of step() has gone stale (f110_launch_epoch)."""
        e = C.c_int64(0)
        _lib.check(self.lib.f110_launch_epoch(self._h, C.byref(e)))
        return e.value

    # ------------------------------------------------------------------ measurement aid
    def profile_begin(self, max_launches, every=1):
        """hipEvent pairs on the scan dispatch of every `every`-th step from now on (f110_profile_begin / _every)."""
        _lib.check(self.lib.f110_profile_every(self._h, int(every)))
        _lib.check(self.lib.f110_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        ms, n = C.c_double(0), C.c_int32(0)
        _lib.check(self.lib.f110_profile_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            torch.cuda.synchronize(self.device)
            flags = self.device_errors() if os.environ.get('F110_CHECK_DEVICE_ERRORS') == '1' else 0
            self.lib.f110_destroy(self._h)
            self._h = None
            if flags:
                # (set for the run of the test suite against the bounds-checked build: no handle may end its life with a
                # device-side complaint nobody read)
                raise RuntimeError('f110 device error word 0x%x at close (bit 0: noise row outside the table, bit 1: an index '
                                   'check of the bounds-checked build failed, bits 8..: which table)' % flags)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrackSet(object):
    """K racelines [M_k,3] = (x, y, speed) packed for f110_pure_pursuit_tracks: waypoints back to back on the device,
    their row offsets (device + host), and the workspace of block bounding boxes (filled by the first call)."""

    def __init__(self, racelines, device):
        lib = _lib.load()
        arrs = []
        for w in racelines:
            w = w.detach().cpu().numpy() if torch.is_tensor(w) else np.asarray(w)
            arrs.append(np.ascontiguousarray(w, dtype=np.float64).reshape(-1, 3))
        if not arrs:
            raise ValueError('TrackSet needs at least one raceline')
        self.K = len(arrs)
        self.offsets = np.ascontiguousarray(np.concatenate([[0], np.cumsum([a.shape[0] for a in arrs])]), dtype=np.int32)
        self.device = torch.device(device)
        self.waypoints = torch.as_tensor(np.concatenate(arrs, axis=0), device=self.device)
        self.offsets_dev = torch.as_tensor(self.offsets, device=self.device)
        n_ws = int(lib.f110_pure_pursuit_workspace(int(self.offsets[-1]), self.K))
        self.workspace = torch.zeros((max(n_ws, 1),), dtype=torch.float64, device=self.device)
        self.boxes_valid = False


def edt_squared(free_mask):
    """Host exact squared EDT (the integer kernel of f110_set_map_occupancy)."""
    lib = _lib.load()
    free_mask = np.ascontiguousarray(free_mask, dtype=np.uint8)
    out = np.empty(free_mask.shape, dtype=np.uint32)
    _lib.check(lib.f110_edt_squared(_np_ptr(free_mask), free_mask.shape[0], free_mask.shape[1], _np_ptr(out)))
    return out


def check_done(poses, start_poses, start_rot, current_time, collisions, near_start, toggles, lap_times, ego_idx=0):
    """F110Env._check_done (f110_env.py:202-244) through f110_check_done, for n envs of A cars.  Device tensors:
    poses / start_poses [n,A,3] f64, start_rot [n,4] f64, current_time [n] f64, collisions [n,A] u8; near_start
    [n,A] u8, toggles [n,A] i32 and lap_times [n,A] f64 are updated IN PLACE.  Returns (lap_counts [n,A] i32,
    done [n] bool, checkpoint_done [n,A] bool)."""
    lib = _lib.load()
    n, A = poses.shape[0], poses.shape[1]
    dev = poses.device
    for t, dt in ((poses, torch.float64), (start_poses, torch.float64), (start_rot, torch.float64),
                  (current_time, torch.float64), (collisions, torch.uint8), (near_start, torch.uint8),
                  (toggles, torch.int32), (lap_times, torch.float64)):
        if t.dtype != dt or not t.is_contiguous() or t.device != dev:
            raise ValueError('check_done: tensors must be contiguous %s on %s' % (dt, dev))
    lap_counts = torch.empty((n, A), dtype=torch.int32, device=dev)
    done = torch.empty((n,), dtype=torch.bool, device=dev)
    ckpt = torch.empty((n, A), dtype=torch.bool, device=dev)
    with torch.cuda.device(dev):  # a NULL handle launches on the calling thread's current device
        _lib.check(lib.f110_check_done(None, _ptr(poses), _ptr(start_poses), _ptr(start_rot), _ptr(current_time),
                                       _ptr(collisions), n, A, int(ego_idx), _ptr(near_start), _ptr(toggles),
                                       _ptr(lap_counts), _ptr(lap_times), _ptr(done), _ptr(ckpt),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return lap_counts, done, ckpt
