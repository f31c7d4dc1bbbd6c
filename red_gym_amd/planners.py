"""Pure-pursuit waypoint follower: the caller on the other side of F110Env.step
(reference: examples/waypoint_follow.py:15-217).  NumPy restatement used by
examples/waypoint_follow.py and as the checker of the batched HIP planner
(SURVEY 8(f-1))."""
import numpy as np


def nearest_point_on_trajectory(point, trajectory):
    """waypoint_follow.py:16-47"""
    diffs = trajectory[1:, :] - trajectory[:-1, :]
    l2s = diffs[:, 0] ** 2 + diffs[:, 1] ** 2
    rel = point - trajectory[:-1, :]
    dots = rel[:, 0] * diffs[:, 0] + rel[:, 1] * diffs[:, 1]
    t = dots / l2s
    t[t < 0.0] = 0.0
    t[t > 1.0] = 1.0
    projections = trajectory[:-1, :] + (t * diffs.T).T
    temp = point - projections
    dists = np.sqrt(temp[:, 0] * temp[:, 0] + temp[:, 1] * temp[:, 1])
    i = int(np.argmin(dists))
    return projections[i], dists[i], t[i], i


def first_point_on_trajectory_intersecting_circle(point, radius, trajectory, t=0.0, wrap=False):
    """waypoint_follow.py:49-129"""
    start_i = int(t)
    start_t = t % 1.0
    first_t = first_i = first_p = None
    n = trajectory.shape[0]

    def seg(i0, i1):
        start = trajectory[i0, :]
        end = trajectory[i1, :] + 1e-6
        V = end - start
        a = V[0] * V[0] + V[1] * V[1]
        sp = start - point
        b = 2.0 * (V[0] * sp[0] + V[1] * sp[1])
        c = (start[0] * start[0] + start[1] * start[1]) + (point[0] * point[0] + point[1] * point[1]) \
            - 2.0 * (start[0] * point[0] + start[1] * point[1]) - radius * radius
        disc = b * b - 4 * a * c
        if disc < 0:
            return None
        disc = np.sqrt(disc)
        return start, V, (-b - disc) / (2.0 * a), (-b + disc) / (2.0 * a)

    for i in range(start_i, n - 1):
        r = seg(i, i + 1)
        if r is None:
            continue
        start, V, t1, t2 = r
        if i == start_i:
            if 0.0 <= t1 <= 1.0 and t1 >= start_t:
                first_t, first_i, first_p = t1, i, start + t1 * V
                break
            if 0.0 <= t2 <= 1.0 and t2 >= start_t:
                first_t, first_i, first_p = t2, i, start + t2 * V
                break
        elif 0.0 <= t1 <= 1.0:
            first_t, first_i, first_p = t1, i, start + t1 * V
            break
        elif 0.0 <= t2 <= 1.0:
            first_t, first_i, first_p = t2, i, start + t2 * V
            break
    if wrap and first_p is None:
        for i in range(-1, start_i):
            r = seg(i % n, (i + 1) % n)
            if r is None:
                continue
            start, V, t1, t2 = r
            if 0.0 <= t1 <= 1.0:
                first_t, first_i, first_p = t1, i, start + t1 * V
                break
            elif 0.0 <= t2 <= 1.0:
                first_t, first_i, first_p = t2, i, start + t2 * V
                break
    return first_p, first_i, first_t


def get_actuation(pose_theta, lookahead_point, position, lookahead_distance, wheelbase):
    """waypoint_follow.py:131-144"""
    d = lookahead_point[0:2] - position
    waypoint_y = np.sin(-pose_theta) * d[0] + np.cos(-pose_theta) * d[1]
    speed = lookahead_point[2]
    if np.abs(waypoint_y) < 1e-6:
        return speed, 0.
    radius = 1 / (2.0 * waypoint_y / lookahead_distance ** 2)
    steering_angle = np.arctan(wheelbase / radius)
    return speed, steering_angle


class PurePursuitPlanner(object):
    """waypoint_follow.py:146-217.  conf needs wpt_path, wpt_delim, wpt_rowskip, wpt_xind,
    wpt_yind, wpt_vind."""

    def __init__(self, conf, wb):
        self.wheelbase = wb
        self.conf = conf
        self.waypoints = np.loadtxt(conf.wpt_path, delimiter=conf.wpt_delim, skiprows=conf.wpt_rowskip)
        self.max_reacquire = 20.
        self._wpts = np.ascontiguousarray(np.vstack((self.waypoints[:, conf.wpt_xind], self.waypoints[:, conf.wpt_yind])).T)

    def render_waypoints(self, *args, **kwargs):
        pass  # drawing belongs to the pyglet renderer, which is out of scope

    def _get_current_waypoint(self, lookahead_distance, position):
        wpts = self._wpts
        nearest_point, nearest_dist, t, i = nearest_point_on_trajectory(position, wpts)
        if nearest_dist < lookahead_distance:
            lookahead_point, i2, t2 = first_point_on_trajectory_intersecting_circle(position, lookahead_distance, wpts,
                                                                                    i + t, wrap=True)
            if i2 is None:
                return None
            current_waypoint = np.empty((3,))
            current_waypoint[0:2] = wpts[i2, :]
            current_waypoint[2] = self.waypoints[i, self.conf.wpt_vind]
            return current_waypoint
        elif nearest_dist < self.max_reacquire:
            return np.append(wpts[i, :], self.waypoints[i, self.conf.wpt_vind])
        return None

    def plan(self, pose_x, pose_y, pose_theta, lookahead_distance, vgain):
        position = np.array([pose_x, pose_y])
        lookahead_point = self._get_current_waypoint(lookahead_distance, position)
        if lookahead_point is None:
            return 4.0, 0.0
        speed, steering_angle = get_actuation(pose_theta, lookahead_point, position, lookahead_distance, self.wheelbase)
        return vgain * speed, steering_angle
