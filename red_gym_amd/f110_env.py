"""F110Env: the reference's single-env Gym API (gym/f110_gym/envs/f110_env.py)
served by the batched HIP step path with a batch of one.

    env = F110Env(map=..., map_ext='.png', num_agents=1, timestep=0.01, integrator=Integrator.RK4)
    obs, reward, done, info = env.reset(np.array([[x, y, theta]]))
    obs, reward, done, info = env.step(np.array([[steer, speed]]))

obs has the reference's keys and container types (base_classes.py:587-603,
f110_env.py:277-278): lists of Python floats, 'scans' a list of float64 arrays.
"""
import numpy as np
import torch

try:  # gym is optional: only the base class and registration come from it
    import gym
    _Base = gym.Env
except Exception:  # pragma: no cover - gym is absent from this image
    gym = None
    _Base = object

from .base_classes import Integrator
from .engine import DEFAULT_PARAMS
from .vec_env import F110VecEnv


class F110Env(_Base):
    metadata = {'render.modes': ['human', 'human_fast']}

    # class-level like the reference (f110_env.py:96-98)
    renderer = None
    current_obs = None
    render_callbacks = []

    def __init__(self, **kwargs):
        # kwargs extraction with the reference's defaults (f110_env.py:100-157); unknown
        # keywords are ignored exactly as there
        self.seed = kwargs.get('seed', 12345)
        self.map_name = kwargs.get('map')  # absent: packaged vegas (f110_env.py:117-118)
        self.map_ext = kwargs.get('map_ext', '.png')
        self.params = kwargs.get('params', dict(DEFAULT_PARAMS))
        self.num_agents = kwargs.get('num_agents', 2)
        self.timestep = kwargs.get('timestep', 0.01)
        self.ego_idx = kwargs.get('ego_idx', 0)
        self.integrator = kwargs.get('integrator', Integrator.RK4)
        self.sim_car_fov = kwargs.get('fov', 2 * np.pi)
        self.start_thresh = 0.5
        self._vec = F110VecEnv(1, map=self.map_name, map_ext=self.map_ext, params=self.params,
                               num_agents=self.num_agents, timestep=self.timestep, ego_idx=self.ego_idx,
                               integrator=self.integrator, fov=self.sim_car_fov, seed=self.seed,
                               device=kwargs.get('device', 0), autoreset=False, keep_f64_scans=True)
        self.map_path = self._vec.map_path
        self.poses_x, self.poses_y, self.poses_theta = [], [], []
        self.collisions = np.zeros((self.num_agents,))
        self.lap_times = np.zeros((self.num_agents,))
        self.lap_counts = np.zeros((self.num_agents,))
        self.current_time = 0.0
        self.toggle_list = np.zeros((self.num_agents,))
        self.render_obs = None

    def _collect(self):
        A = self.num_agents
        # ONE device -> host hop per step: a gather kernel packs every field of the env into one fp64 row (counters, flags
        # and fp64 times are all exactly representable) in a pinned buffer
        small = self._vec.eng.pack_env(0)
        scans = small[11 * A + 2:].reshape(A, -1)
        st = small[:7 * A].reshape(A, 7).copy()
        o = 7 * A
        self.collisions = small[o:o + A].copy()
        self.lap_times = small[o + A:o + 2 * A].copy()
        self.lap_counts = small[o + 2 * A:o + 3 * A].copy()
        self.toggle_list = small[o + 3 * A:o + 4 * A].copy()
        self.current_time = float(small[o + 4 * A])
        done = bool(small[o + 4 * A + 1])
        obs = {'ego_idx': self.ego_idx,
               'scans': [scans[i].copy() for i in range(self.num_agents)],
               'poses_x': [float(st[i, 0]) for i in range(self.num_agents)],
               'poses_y': [float(st[i, 1]) for i in range(self.num_agents)],
               'poses_theta': [float(st[i, 4]) for i in range(self.num_agents)],
               'linear_vels_x': [float(st[i, 3]) for i in range(self.num_agents)],
               'linear_vels_y': [0. for _ in range(self.num_agents)],
               'ang_vels_z': [float(st[i, 5]) for i in range(self.num_agents)],
               'collisions': self.collisions,
               'lap_times': self.lap_times,
               'lap_counts': self.lap_counts}
        F110Env.current_obs = obs
        self.poses_x, self.poses_y, self.poses_theta = obs['poses_x'], obs['poses_y'], obs['poses_theta']
        self.render_obs = {k: obs[k] for k in ('ego_idx', 'poses_x', 'poses_y', 'poses_theta', 'lap_times', 'lap_counts')}
        info = {'checkpoint_done': self.toggle_list >= 4}
        return obs, self.timestep, done, info

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        self._vec.step(action.reshape(1, self.num_agents, 2))
        return self._collect()

    def reset(self, poses):
        poses = np.asarray(poses, dtype=np.float64)
        if poses.ndim != 2 or poses.shape[0] != self.num_agents:
            raise ValueError('Number of poses for reset does not match number of agents.')
        self._vec.reset(poses.reshape(1, self.num_agents, 3))
        return self._collect()

    def update_map(self, map_path, map_ext):
        self._vec.update_map(map_path, map_ext)

    def update_params(self, params, index=-1):
        self._vec.update_params(params, index)
        self.params = params

    def add_render_callback(self, callback_func):
        F110Env.render_callbacks.append(callback_func)

    def render(self, mode='human'):
        """The pyglet/OpenGL window of the reference (rendering.py) is out of scope.  The call
        is accepted (callers invoke it every step); with `env.start_recording(max_steps)` it
        logs the frame the window would have shown (poses, laps) for `env.save_recording()`."""
        assert mode in ['human', 'human_fast']
        if getattr(self, '_recorder', None) is not None and self._recorder.t < self._recorder.max_steps:
            self._recorder.record()

    def start_recording(self, max_steps, with_scans=False):
        from .recorder import TrajectoryRecorder
        self._recorder = TrajectoryRecorder(self._vec, max_steps, with_scans=with_scans)
        return self._recorder

    def save_recording(self, path):
        return self._recorder.save(path)

    def close(self):
        self._vec.close()
