"""Synthetic benchmark workload of SURVEY.md section 8(d): spawn poses along the
example_map raceline and the random action distribution."""
import os

import numpy as np

from .maps import ASSETS, load_waypoints

EXAMPLE_MAP = os.path.join(ASSETS, 'example_map')  # '<...>/example_map' + '.yaml' / '.png'
RACELINE = os.path.join(ASSETS, 'example_waypoints.csv')


def spawn_poses(num_envs, num_agents, rank=0):
    """env e, agent a: raceline row k=(97e + 11a) mod 783 (cols x_m=1, y_m=2, psi_rad=3),
    pose (x_k, y_k, psi_k + pi/2) + jitter N(0, 0.2^2) in x, y and yaw (clipped +-0.5) from
    default_rng(2025 + rank); with 2+ agents the others sit 1.5 m steps behind the first
    along the raceline (so GJK / opponent ray-cast see real neighbours)."""
    rl = load_waypoints(RACELINE)
    n = rl.shape[0]
    rng = np.random.default_rng(2025 + rank)
    poses = np.zeros((num_envs, num_agents, 3))
    e = np.arange(num_envs)
    step = rl[1, 0] - rl[0, 0]  # arc length between raceline rows (~0.2 m)
    back = int(round(1.5 / step))
    for a in range(num_agents):
        k = (97 * e + 11 * 0) % n if num_agents > 1 else (97 * e + 11 * a) % n
        k = (k - back * a) % n
        poses[:, a, 0] = rl[k, 1]
        poses[:, a, 1] = rl[k, 2]
        poses[:, a, 2] = rl[k, 3] + np.pi / 2
    poses[:, :, 0] += rng.normal(0, 0.2, (num_envs, num_agents))
    poses[:, :, 1] += rng.normal(0, 0.2, (num_envs, num_agents))
    poses[:, :, 2] += np.clip(rng.normal(0, 0.2, (num_envs, num_agents)), -0.5, 0.5)
    return poses


def action_pool(pool, num_envs, num_agents, rank=0, s_max=0.4189, v_hi=8.0):
    """steer ~ U(-s_max, s_max), speed ~ U(0, 8) (pattern of the reference's random-action
    recorder f1tenth_gym/examples/lidar.py:205-206), seeded 777 + rank."""
    rng = np.random.default_rng(777 + rank)
    act = np.empty((pool, num_envs, num_agents, 2))
    act[..., 0] = rng.uniform(-s_max, s_max, (pool, num_envs, num_agents))
    act[..., 1] = rng.uniform(0.0, v_hi, (pool, num_envs, num_agents))
    return act
