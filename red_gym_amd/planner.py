"""PurePursuitPlanner with the reference's interface (examples/waypoint_follow.py:146-217: constructed from
the config namespace and the wheelbase, `plan(x, y, theta, lookahead, vgain) -> (speed, steer)`), planned by
the batched HIP kernel `f110_pure_pursuit` (csrc/f110_planner.h) on a batch of one.  For thousands of cars call
`F110VecEnv.pure_pursuit`, which feeds the same kernel from the state tensor without leaving the GPU."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class PurePursuitPlanner(object):
    def __init__(self, conf, wb, device=0):
        self.lib = _lib.load()  # raises without the HIP library: there is no host planner in the product
        self.wheelbase, self.conf, self.max_reacquire = wb, conf, 20.
        self.waypoints = np.loadtxt(conf.wpt_path, delimiter=conf.wpt_delim, skiprows=conf.wpt_rowskip)
        self.device = torch.device('cuda', int(device))
        xyv = np.ascontiguousarray(self.waypoints[:, [conf.wpt_xind, conf.wpt_yind, conf.wpt_vind]], dtype=np.float64)
        self._wp = torch.as_tensor(xyv, device=self.device)
        self._state = torch.zeros((1, 7), dtype=torch.float64, device=self.device)
        self._pose = torch.zeros((3,), dtype=torch.float64).pin_memory()
        self._act = torch.zeros((1, 2), dtype=torch.float64, device=self.device)
        self._idx = torch.tensor([0, 1, 4], device=self.device)

    def render_waypoints(self, *args, **kwargs):
        pass  # drawing belongs to the pyglet renderer (rendering.py), which is out of scope

    def plan(self, pose_x, pose_y, pose_theta, lookahead_distance, vgain):
        self._pose[0], self._pose[1], self._pose[2] = float(pose_x), float(pose_y), float(pose_theta)
        self._state[0, self._idx] = self._pose.to(self.device, non_blocking=True)
        with torch.cuda.device(self.device):  # h = NULL: the kernel goes to the current device, which must be ours
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self.lib.f110_pure_pursuit(None, C.c_void_p(self._wp.data_ptr()), self._wp.shape[0],
                                                  float(lookahead_distance), float(vgain), float(self.wheelbase),
                                                  float(self.max_reacquire), C.c_void_p(self._state.data_ptr()), 1,
                                                  C.c_void_p(self._act.data_ptr()), st))
        steer, speed = self._act[0].tolist()  # one device -> host hop per plan (the reference plans on the host)
        return speed, steer
