"""CHECKER for the batched HIP pure-pursuit planner -- test infrastructure, never imported by
the product (red_gym_amd/, examples/, bench.py's timed region).

NumPy restatement of the reference's waypoint follower, examples/waypoint_follow.py:15-217, in a vectorised
form of our own: the reference walks the raceline segment by segment in a Numba loop; here every candidate
segment's circle intersection is computed at once and the first admissible one is picked, which gives the
same segment index (the only thing the planner uses from the search).  Pinned by tests/golden/g8_env.npz:
the actions the reference's own planner produced in its 3 329-step closed loop (<= 1e-12,
tests/test_host_cpu.py::test_planner_reproduces_reference_actions).
"""
import numpy as np


class Raceline(object):
    """Waypoint polyline [M,2] with the per-segment quantities of nearest_point_on_trajectory
    (waypoint_follow.py:16-47) precomputed."""

    def __init__(self, xy):
        self.xy = np.ascontiguousarray(xy, dtype=np.float64)
        self.seg = self.xy[1:] - self.xy[:-1]
        self.len2 = self.seg[:, 0] ** 2 + self.seg[:, 1] ** 2

    def nearest(self, p):
        """waypoint_follow.py:16-47 -> (distance, t in [0,1] on segment i, i); first minimum wins."""
        rel = p - self.xy[:-1]
        t = np.clip((rel[:, 0] * self.seg[:, 0] + rel[:, 1] * self.seg[:, 1]) / self.len2, 0.0, 1.0)
        off = p - (self.xy[:-1] + t[:, None] * self.seg)
        d = np.sqrt(off[:, 0] * off[:, 0] + off[:, 1] * off[:, 1])
        i = int(np.argmin(d))
        return d[i], t[i], i

    def circle_hit(self, p, radius, progress):
        """Index of the first segment, from `progress` = i + t onwards and then wrapping from segment -1
        (last point -> first point) up to it, that the circle (p, radius) intersects at a parameter in [0, 1]
        (>= t on the starting segment): waypoint_follow.py:49-129 with wrap=True.  None if there is none.
        The index can be -1, which the caller uses as a Python index (= last waypoint), as the reference does."""
        n = self.xy.shape[0]
        i0, t0 = int(progress), progress % 1.0
        order = np.concatenate([np.arange(i0, n - 1), np.arange(-1, i0)])  # forward pass, then the wrap pass
        a_pt = self.xy[order % n]
        v = (self.xy[(order + 1) % n] + 1e-6) - a_pt                         # end = trajectory[i+1] + 1e-6 (:73)
        qa = v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]
        qb = 2.0 * (v[:, 0] * (a_pt[:, 0] - p[0]) + v[:, 1] * (a_pt[:, 1] - p[1]))
        qc = (a_pt[:, 0] * a_pt[:, 0] + a_pt[:, 1] * a_pt[:, 1]) + (p[0] * p[0] + p[1] * p[1]) \
            - 2.0 * (a_pt[:, 0] * p[0] + a_pt[:, 1] * p[1]) - radius * radius
        disc = qb * qb - 4 * qa * qc
        ok = disc >= 0
        root = np.sqrt(np.where(ok, disc, 0.0))
        t1, t2 = (-qb - root) / (2.0 * qa), (-qb + root) / (2.0 * qa)
        lo = np.zeros(order.shape[0])
        if n - 1 > i0:
            lo[0] = t0                                                       # only the starting segment (:85-97)
        hit = ok & (((t1 >= 0.0) & (t1 <= 1.0) & (t1 >= lo)) | ((t2 >= 0.0) & (t2 <= 1.0) & (t2 >= lo)))
        k = np.flatnonzero(hit)
        return int(order[k[0]]) if k.size else None


def get_actuation(pose_theta, target, position, lookahead_distance, wheelbase):
    """waypoint_follow.py:131-144 -> (speed, steering angle)"""
    dx, dy = target[0] - position[0], target[1] - position[1]
    lateral = np.sin(-pose_theta) * dx + np.cos(-pose_theta) * dy
    if np.abs(lateral) < 1e-6:
        return target[2], 0.
    radius = 1 / (2.0 * lateral / lookahead_distance ** 2)
    return target[2], np.arctan(wheelbase / radius)


class PurePursuitPlanner(object):
    """waypoint_follow.py:146-217.  conf: wpt_path, wpt_delim, wpt_rowskip, wpt_xind, wpt_yind, wpt_vind."""

    def __init__(self, conf, wb):
        self.wheelbase, self.conf, self.max_reacquire = wb, conf, 20.
        self.waypoints = np.loadtxt(conf.wpt_path, delimiter=conf.wpt_delim, skiprows=conf.wpt_rowskip)
        self.line = Raceline(self.waypoints[:, [conf.wpt_xind, conf.wpt_yind]])
        self.speeds = self.waypoints[:, conf.wpt_vind]

    def target(self, lookahead_distance, position):
        """_get_current_waypoint (:183-204): (x, y, speed) to steer at, or None."""
        dist, t, i = self.line.nearest(position)
        if dist < lookahead_distance:
            j = self.line.circle_hit(position, lookahead_distance, i + t)
            if j is None:
                return None
            return np.array([self.line.xy[j, 0], self.line.xy[j, 1], self.speeds[i]])
        if dist < self.max_reacquire:
            return np.array([self.line.xy[i, 0], self.line.xy[i, 1], self.speeds[i]])
        return None

    def plan(self, pose_x, pose_y, pose_theta, lookahead_distance, vgain):
        position = np.array([pose_x, pose_y])
        tgt = self.target(lookahead_distance, position)
        if tgt is None:
            return 4.0, 0.0                                                  # :211-212
        speed, steer = get_actuation(pose_theta, tgt, position, lookahead_distance, self.wheelbase)
        return vgain * speed, steer
