"""Names the reference exposes from f110_gym.envs.base_classes that callers import
(examples/waypoint_follow.py:2 `from f110_gym.envs.base_classes import Integrator`)."""
from enum import Enum


class Integrator(Enum):
    """base_classes.py:40-42"""
    RK4 = 1
    Euler = 2


def integrator_code(integrator):
    """Accepts this enum, the reference's own enum (same names), or 1/2.
    An unknown integrator is the reference's SyntaxError (base_classes.py:396)."""
    name = getattr(integrator, 'name', None)
    if name == 'RK4' or integrator == 1:
        return 1
    if name == 'Euler' or integrator == 2:
        return 2
    raise SyntaxError('Invalid Integrator Specified. Provided %s. Please choose RK4 or Euler'
                      % (name if name is not None else integrator))


class RaceCar(object):
    """View of one car of a Simulator (base_classes.py:44): `.state` is the 7-vector
    [x, y, steer, v, yaw, yaw_rate, slip], `.in_collision` the iTTC flag."""

    def __init__(self, sim, index):
        self._sim, self._i = sim, index

    @property
    def state(self):
        return self._sim._vec.eng.t['state'][0, self._i].cpu().numpy()

    @property
    def in_collision(self):
        return bool(self._sim._vec.eng.t['in_collision'][0, self._i].item())


class Simulator(object):
    """Reference signature (base_classes.py:459): Simulator(params, num_agents, seed, fov,
    time_step=0.01, ego_idx=0, integrator=Integrator.RK4); set_map / reset / step /
    update_params with the same observation dict (:587-603), served by a batch of one."""

    def __init__(self, params, num_agents, seed, fov, time_step=0.01, ego_idx=0, integrator=Integrator.RK4):
        self.params, self.num_agents, self.seed, self.fov = params, num_agents, seed, fov
        self.time_step, self.ego_idx, self.integrator = time_step, ego_idx, integrator
        self._vec = None
        self.agents = [RaceCar(self, i) for i in range(num_agents)]
        import numpy as np
        self.collisions = np.zeros((num_agents,))
        self.collision_idx = -1 * np.ones((num_agents,))
        self.agent_poses = np.empty((num_agents, 3))

    def set_map(self, map_path, map_ext):
        import os
        from .vec_env import F110VecEnv
        if self._vec is None:
            self._vec = F110VecEnv(1, map=os.path.splitext(map_path)[0], map_ext=map_ext, params=self.params,
                                   num_agents=self.num_agents, timestep=self.time_step, ego_idx=self.ego_idx,
                                   integrator=self.integrator, fov=self.fov, seed=self.seed, autoreset=False,
                                   keep_f64_scans=True)
            for prm, idx in getattr(self, '_pending_params', []):
                self._vec.update_params(prm, idx)
            self._pending_params = []
        else:
            self._vec.update_map(map_path, map_ext)

    def update_params(self, params, agent_idx=-1):
        if agent_idx >= self.num_agents:
            raise IndexError('Index given is out of bounds for list of agents.')
        if self._vec is not None:
            self._vec.update_params(params, agent_idx)
        else:
            self._pending_params = getattr(self, '_pending_params', []) + [(params, agent_idx)]

    def _need_map(self):
        if self._vec is None:
            raise ValueError('Map is not set for scan simulator.')

    def reset(self, poses):
        """base_classes.py:607-623: places the cars; unlike F110Env.reset no step is taken
        here, so the zero-action step that the batched reset performs is undone by
        re-arming the state (the reference's Simulator.reset + first step(0) == our reset)."""
        import numpy as np
        self._need_map()
        poses = np.asarray(poses, dtype=np.float64)
        if poses.shape[0] != self.num_agents:
            raise ValueError('Number of poses for reset does not match number of agents.')
        t = self._vec.eng.t
        import torch
        t['spawn'][0] = torch.as_tensor(poses, device=t['spawn'].device)
        t['pending_reset'][0] = 1
        self._armed = True

    def step(self, control_inputs):
        import numpy as np
        import torch
        self._need_map()
        eng = self._vec.eng
        t = eng.t
        control_inputs = np.asarray(control_inputs, dtype=np.float64).reshape(1, self.num_agents, 2)
        if getattr(self, '_armed', False):
            # RaceCar.reset state (zero state at the pose, empty steer FIFO, noise restarted)
            t['state'].zero_()
            t['state'][0, :, 0] = t['spawn'][0, :, 0]
            t['state'][0, :, 1] = t['spawn'][0, :, 1]
            t['state'][0, :, 4] = t['spawn'][0, :, 2]
            t['steer_buf'].zero_(); t['steer_cnt'].zero_(); t['noise_step'].zero_()
            t['pending_reset'][0] = 0
            self._armed = False
        eng.step(torch.as_tensor(control_inputs, device=eng.device))
        st = t['state'][0].cpu().numpy()
        scans = t['scans_f64'][0].cpu().numpy()
        self.collisions = t['collisions'][0].cpu().numpy().astype(np.float64)
        self.collision_idx = t['collision_idx'][0].cpu().numpy().astype(np.float64)
        self.agent_poses = t['pose_snap'][0].cpu().numpy().copy()
        A = self.num_agents
        return {'ego_idx': self.ego_idx, 'scans': [scans[i].copy() for i in range(A)],
                'poses_x': [float(st[i, 0]) for i in range(A)], 'poses_y': [float(st[i, 1]) for i in range(A)],
                'poses_theta': [float(st[i, 4]) for i in range(A)],
                'linear_vels_x': [float(st[i, 3]) for i in range(A)], 'linear_vels_y': [0. for _ in range(A)],
                'ang_vels_z': [float(st[i, 5]) for i in range(A)], 'collisions': self.collisions}
