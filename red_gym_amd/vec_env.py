"""F110VecEnv: B independent F1TENTH envs stepped in lock-step on one MI355X.

Same constructor keywords, observation keys and step order as the reference's
F110Env (gym/f110_gym/envs/f110_env.py:100-157, :261-347), with a leading batch
dimension and torch tensors instead of Python lists.  Nothing is computed here:
reset/step enqueue the HIP kernels of libf110_hip.so on the current stream.
"""
import os

import numpy as np
import torch

from .base_classes import Integrator
from .engine import DEFAULT_PARAMS, Engine
from .maps import BUILTIN_MAPS, DEFAULT_MAP, builtin_map_yaml


def resolve_map_path(map_name):
    """f110_env.py:106-118: 'berlin' / 'skirk' / 'levine' are packaged maps, any other name (an explicit 'vegas'
    included) means '<map>.yaml' relative to the working directory, and only an ABSENT `map` keyword (None
    here) selects the packaged vegas."""
    if map_name is None:
        return builtin_map_yaml(DEFAULT_MAP)
    if map_name in BUILTIN_MAPS:
        return builtin_map_yaml(map_name)
    return map_name + '.yaml'


class F110VecEnv(object):
    def __init__(self, num_envs, map=None, map_ext='.png', params=None, num_agents=2, timestep=0.01,
                 ego_idx=0, integrator=Integrator.RK4, fov=2 * np.pi, seed=12345, device=0, autoreset=True,
                 num_beams=1080, noise_std=0.01, noise_steps=0, keep_f64_scans=False, count_lookups=False,
                 noise_source='device', **_ignored):
        """The reference's constructor keywords (f110_env.py:100-157) plus the batch: `params` and `seed` may each be ONE
        value for every env, or a sequence of num_envs values -- env e is then what `F110Env(params=params[e],
        seed=seed[e])` would be (equal values share a slot on the device; more than 64 distinct seeds switch the device noise to
        one generator per env, `noise_source='per_env'`).  The scan's beam tables and side distances are built ONCE, from env 0's
        params -- what reference envs created in one process share through RaceCar's class-level statics (base_classes.py:116-156),
        not what independently started processes would have: an env whose `width` / `lf` / `lr` differ still uses env 0's side
        distances for its iTTC test."""
        self.num_envs, self.num_agents = int(num_envs), int(num_agents)
        self.map_name, self.map_ext = map, map_ext
        self.map_path = resolve_map_path(map)
        self.params = dict(DEFAULT_PARAMS if params is None else (params if isinstance(params, dict) else params[0]))
        self.timestep, self.ego_idx, self.seed = timestep, ego_idx, seed
        self.eng = Engine(num_envs=num_envs, num_agents=num_agents, params=self.params if params is None or isinstance(params, dict) else params,
                          seed=seed, fov=fov,
                          timestep=timestep, integrator=integrator, ego_idx=ego_idx, num_beams=num_beams,
                          device=device, autoreset=autoreset, noise_std=noise_std, noise_steps=noise_steps,
                          keep_f64_scans=keep_f64_scans, count_lookups=count_lookups, noise_source=noise_source)
        self.eng.set_map(self.map_path, self.map_ext)
        self.device = self.eng.device
        t = self.eng.t
        st = t['state']
        # observation views (no copies): obs keys of base_classes.py:587-603 + f110_env.py:277-278
        self._obs = {
            'ego_idx': ego_idx,
            'scans': t['scans'],
            'poses_x': st[..., 0], 'poses_y': st[..., 1], 'poses_theta': st[..., 4],
            'linear_vels_x': st[..., 3],
            'linear_vels_y': torch.zeros((self.num_envs, self.num_agents), dtype=torch.float64, device=self.device),
            'ang_vels_z': st[..., 5],
            'collisions': t['collisions'],
            'lap_times': t['lap_times'], 'lap_counts': t['lap_counts'],
        }
        if t['scans_f64'] is not None:
            self._obs['scans_f64'] = t['scans_f64']
        self._reward = torch.full((self.num_envs,), float(timestep), dtype=torch.float64, device=self.device)

    def _result(self):
        t = self.eng.t
        done = t['done']  # bool tensor written by env_kernel (no per-step torch kernels here)
        info = {'checkpoint_done': t['checkpoint_done'], 'collision_idx': t['collision_idx'],
                'current_time': t['current_time'], 'toggles': t['toggles']}
        return self._obs, self._reward, done, info

    def _as_dev(self, a, last):
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))
        a = a.to(device=self.device, dtype=torch.float64)
        if a.dim() == 2 and self.num_envs == 1:
            a = a.unsqueeze(0)
        if tuple(a.shape) != (self.num_envs, self.num_agents, last):
            raise ValueError('expected shape (%d, %d, %d), got %s' % (self.num_envs, self.num_agents, last, tuple(a.shape)))
        return a.contiguous()

    def reset(self, poses, mask=None):
        """poses [B,A,3] (x, y, yaw).  Returns (obs, reward, done, info) like the
        reference's reset (which performs one zero-action step, f110_env.py:335-347).
        mask [B]: reset only those envs (others keep running untouched)."""
        try:
            poses = self._as_dev(poses, 3)
        except ValueError:
            raise ValueError('Number of poses for reset does not match number of agents.')
        self.eng.reset(poses, mask)
        return self._result()

    def step(self, actions):
        """actions [B,A,2] = (steer, speed) per car (f110_env.py:261-302)."""
        self.eng.step(self._as_dev(actions, 2))
        return self._result()

    # ------------------------------------------------------------------ checkpoint / resume
    _STATE_KEYS = ('state', 'steer_buf', 'steer_cnt', 'noise_step', 'spawn', 'start_rot', 'near_start', 'toggles',
                   'current_time', 'pending_reset', 'collisions', 'collision_idx', 'in_collision', 'lap_counts',
                   'lap_times', 'done', 'checkpoint_done', 'scans')

    def state_dict(self):
        """Everything a step depends on lives in the caller-owned tensors bound to the handle
        (f110_buffers): a copy of them is a complete checkpoint of all B envs."""
        return {k: self.eng.t[k].clone() for k in self._STATE_KEYS if self.eng.t[k] is not None}

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.eng.t[k].copy_(v)
        # the host's upper bound of any car's noise row must cover the restored counters, and the noise table their rows
        self.eng.host_steps_bound = max(self.eng.host_steps_bound, int(self.eng.t['noise_step'].max().item()) + 1)
        self.eng._steps_exact = False
        with torch.cuda.device(self.device):
            self.eng._set_noise_floor(0)

    # ------------------------------------------------------------------ hipGraph replay
    def capture_step(self, policy=None, copies=1):
        """Captures one step (optionally preceded by a device-side policy that fills the
        action buffer, e.g. `lambda env, out: env.eng.pure_pursuit(wp, tlad, vgain, out=out)`)
        into a HIP graph.  f110_step neither allocates nor synchronises, so the three or four
        kernel launches replay from one graph launch; `step_graph()` then costs one host call.
        Returns the static action buffer [B,A,2] to write into when no policy is given (it stays the
        same tensor across re-captures)."""
        if getattr(self, '_g_actions', None) is None:
            self._g_actions = torch.zeros((self.num_envs, self.num_agents, 2), dtype=torch.float64, device=self.device)
        self._g_policy = policy
        self.eng._ensure_noise()
        # `copies` > 1 captures that many identical graphs, replayed in turn (an experiment: two alternating execs
        # replay no faster than one, profiles/r02_graph_vs_eager.txt)
        if getattr(self, '_graphs', None):
            # a replay of the graphs being dropped may still be in flight (f110_set_scan_stages bumps the epoch without
            # synchronising): a graph exec must outlive its last launch
            torch.cuda.current_stream(self.device).synchronize()
        self._graphs, self._g_next = [], 0
        for _ in range(max(1, int(copies))):
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            g = torch.cuda.CUDAGraph()
            self.eng._in_capture = True
            try:
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        if policy is not None:
                            policy(self, self._g_actions.view(-1, 2))
                        self.eng.step(self._g_actions)
            finally:
                self.eng._in_capture = False
            torch.cuda.current_stream(self.device).wait_stream(side)
            self.eng.host_steps_bound -= 1  # the capture executed nothing: undo its host-side step accounting
            self._graphs.append(g)
        self._g_copies = len(self._graphs)
        # a capture freezes the kernel choice and the by-value arguments (noise table address and length, map
        # template flags, env -> map table): it is valid for this launch epoch only
        self._g_epoch = self.eng.launch_epoch()
        return self._g_actions

    def step_graph(self, actions=None):
        """Replays the captured step.  With `actions` they are copied into the static buffer first; cheaper is to
        write into the buffer capture_step returned (or to capture a policy).  If the handle's launch epoch moved
        since the capture (the noise table grew, a map with other template flags was installed, tracks were
        randomised) the step is re-captured first, so a replay never reads a freed table."""
        if actions is not None:
            self._g_actions.copy_(self._as_dev(actions, 2))
        self.eng._ensure_noise()
        if self.eng.launch_epoch() != self._g_epoch:
            self.capture_step(self._g_policy, self._g_copies)
        self._graphs[self._g_next].replay()
        self._g_next = (self._g_next + 1) % len(self._graphs)
        self.eng.host_steps_bound += 1
        return self._result()

    # ------------------------------------------------------------------ hipGraph built by the library
    def build_step_graph(self, how='nodes'):
        """The step as a HIP graph built by the library itself (f110_graph_create: 'nodes' = explicit kernel nodes,
        'capture' = a capture on a private non-blocking stream) instead of a torch capture.  Returns the static action
        buffer [B,A,2]; `step_lib_graph()` replays."""
        import ctypes as C
        from . import _lib
        if getattr(self, '_g_actions', None) is None:
            self._g_actions = torch.zeros((self.num_envs, self.num_agents, 2), dtype=torch.float64, device=self.device)
        self._drop_lib_graph()
        self.eng._ensure_noise()
        g = C.c_void_p()
        _lib.check(self.eng.lib.f110_graph_create(self.eng._h, C.c_void_p(self._g_actions.data_ptr()),
                                                  {'nodes': 0, 'capture': 1}[how], C.byref(g)))
        self._lg, self._lg_how, self._lg_epoch = g, how, self.eng.launch_epoch()
        return self._g_actions

    def _drop_lib_graph(self):
        if getattr(self, '_lg', None):
            torch.cuda.current_stream(self.device).synchronize()  # a graph exec must outlive its last launch
            self.eng.lib.f110_graph_destroy(self._lg)
            self._lg = None

    def lib_graph_info(self, dot_path=None):
        import ctypes as C
        from . import _lib
        n = C.c_int32(0)
        _lib.check(self.eng.lib.f110_graph_info(self._lg, C.byref(n), dot_path.encode() if dot_path else None))
        return n.value

    def step_lib_graph(self, actions=None):
        from . import _lib
        if actions is not None:
            self._g_actions.copy_(self._as_dev(actions, 2))
        self.eng._ensure_noise()
        if self.eng.launch_epoch() != self._lg_epoch:
            self.build_step_graph(self._lg_how)
        with torch.cuda.device(self.device):
            _lib.check(self.eng.lib.f110_graph_launch(self._lg, self.eng._stream()))
        self.eng.host_steps_bound += 1
        return self._result()

    def pure_pursuit(self, waypoints, lookahead, vgain, wheelbase=0.17145 + 0.15875, prepare=True):
        """Batched pure-pursuit actions for the current poses (examples/waypoint_follow.py planner
        on the GPU); waypoints [M,3] = (x, y, speed).  A device tensor that is planned on repeatedly is prepared once
        (Engine.pure_pursuit) and then costs one lane per car."""
        if not torch.is_tensor(waypoints) or waypoints.device != self.device:
            waypoints = torch.as_tensor(np.ascontiguousarray(waypoints, dtype=np.float64), device=self.device)
        return self.eng.pure_pursuit(waypoints, lookahead, vgain, wheelbase, prepare=prepare)

    def raceline_slots(self, waypoint_sets, assign):
        """Packs K racelines [M_k,3] = (x, y, speed) and the raceline of every env (int array [num_envs], e.g. the map
        slots of randomize_tracks) for pure_pursuit_tracks: the planner's counterpart of f110_assign_maps.  Returns
        (TrackSet, int32 device tensor [num_envs * num_agents])."""
        from .engine import TrackSet
        assign = np.asarray(assign)
        if assign.shape != (self.num_envs,) or assign.min() < 0 or assign.max() >= len(waypoint_sets):
            raise ValueError('assign must hold one raceline index (0..%d) per env' % (len(waypoint_sets) - 1))
        ts = TrackSet(waypoint_sets, self.device)
        of_car = torch.as_tensor(np.repeat(assign.astype(np.int32), self.num_agents), device=self.device)
        return ts, of_car

    def pure_pursuit_tracks(self, tracks, track_of_car, lookahead, vgain, wheelbase=0.17145 + 0.15875):
        """Pure-pursuit actions [B,A,2] when every env has its own raceline (raceline_slots): ONE planner launch for
        all tracks, racelines of any length, capturable in a hipGraph together with the step."""
        return self.eng.pure_pursuit_tracks(tracks, track_of_car, lookahead, vgain, wheelbase)

    def pure_pursuit_blocks(self, waypoint_sets, assign, lookahead, vgain, wheelbase=0.17145 + 0.15875):
        """Pure-pursuit actions when blocks of envs drive on different tracks (randomize_tracks): waypoint_sets[k]
        is the raceline [M_k,3] = (x, y, speed) of slot k, assign the int array [num_envs] of slots.  One planner
        launch for all of them (the packed racelines are cached while the same objects are passed)."""
        # (the packed form is rebuilt when another list, other raceline objects or another assignment is passed; a raceline
        # edited IN PLACE is not noticed -- build a new TrackSet with raceline_slots for that)
        key = (tuple((id(w), tuple(getattr(w, 'shape', ()))) for w in waypoint_sets), np.asarray(assign).tobytes())
        if getattr(self, '_pp_tracks_key', None) != key:
            self._pp_tracks = self.raceline_slots(waypoint_sets, assign)
            self._pp_tracks_key = key
        ts, of_car = self._pp_tracks
        return self.eng.pure_pursuit_tracks(ts, of_car, lookahead, vgain, wheelbase)

    def update_params(self, params, index=-1):
        """base_classes.py:507-527: index < 0 updates every agent, otherwise agent `index` of
        every env; IndexError beyond the agent list."""
        self.eng.update_params(params, index)
        if index < 0:
            self.params = dict(params)

    def update_map(self, map_path, map_ext):
        self.eng.set_map(map_path, map_ext)

    def update_map_occupancy(self, free, resolution, orig_x, orig_y, orig_theta=0.0):
        """Installs a map given as an occupancy mask (NumPy array or CUDA uint8 tensor, nonzero = free, row 0 at the
        bottom); with a device tensor nothing but two scalars leaves the GPU."""
        self.eng.set_map_occupancy(free, resolution, orig_x, orig_y, orig_theta)

    def randomize_track(self, seed):
        """Domain randomisation: draws a random closed track (red_gym_amd.trackgen, the reference's
        unittest/random_trackgen.py) on the GPU, installs it as the map of every env of this shard and returns the
        Track (centre-line waypoints [N,3] = x, y, heading in world metres; waypoint 0 is the world origin)."""
        from . import trackgen
        t = trackgen.generate(seed, device=self.device)
        self.eng.set_map_occupancy(t.free, t.resolution, t.orig_x, t.orig_y, 0.0)
        return t

    def randomize_tracks(self, seeds):
        """One random track per seed, installed in map slots 0..K-1, with the envs split into K equal blocks (env e
        drives on track e*K // num_envs) -- what K separate reference envs with their own `map` would be.  Returns
        the Tracks and the int array [num_envs] of slots."""
        from . import trackgen
        seeds = list(seeds)
        tracks = []
        for k, seed in enumerate(seeds):
            t = trackgen.generate(seed, device=self.device)
            self.eng.set_map_occupancy(t.free, t.resolution, t.orig_x, t.orig_y, 0.0, slot=k)
            tracks.append(t)
        assign = (np.arange(self.num_envs) * len(seeds)) // self.num_envs
        self.eng.assign_maps(assign)
        return tracks, assign

    @property
    def state(self):
        return self.eng.t['state']

    def close(self):
        self._drop_lib_graph()
        self.eng.close()
