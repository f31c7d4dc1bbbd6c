"""Config 1 of BASELINE.json: one env, one agent, pure pursuit around example_map -- the
same calling pattern as the reference's examples/waypoint_follow.py:241-287
(gym.make('f110_gym:f110-v0', ...), env.reset(poses), env.step(np.array([[steer, speed]])),
env.render()), served by the HIP step path.

    python examples/waypoint_follow.py
expected: done after 3329 steps, sim time 33.29 s, 2 laps, no collision.
"""
import os
import sys
import time
from argparse import Namespace

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from red_gym_amd import compat  # noqa: E402
compat.install_missing()  # gym / numba / pyglet stand-ins when those packages are absent

import gym  # noqa: E402
from f110_gym.envs.base_classes import Integrator  # noqa: E402
from red_gym_amd.maps import ASSETS  # noqa: E402
from red_gym_amd.planner import PurePursuitPlanner  # noqa: E402


def main():
    work = {'mass': 3.463388126201571, 'lf': 0.15597534362552312, 'tlad': 0.82461887897713965, 'vgain': 1.375}
    with open(os.path.join(ASSETS, 'config_example_map.yaml')) as f:
        conf = Namespace(**yaml.safe_load(f))
    conf.map_path = os.path.join(ASSETS, 'example_map')
    conf.wpt_path = os.path.join(ASSETS, 'example_waypoints.csv')
    planner = PurePursuitPlanner(conf, (0.17145 + 0.15875))

    env = gym.make('f110_gym:f110-v0', map=conf.map_path, map_ext=conf.map_ext, num_agents=1, timestep=0.01,
                   integrator=Integrator.RK4)
    env.add_render_callback(lambda renderer: planner.render_waypoints(renderer))
    obs, step_reward, done, info = env.reset(np.array([[conf.sx, conf.sy, conf.stheta]]))
    env.render()
    laptime, steps = 0.0, 0
    start = time.time()
    while not done:
        speed, steer = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], work['tlad'], work['vgain'])
        obs, step_reward, done, info = env.step(np.array([[steer, speed]]))
        laptime += step_reward
        steps += 1
        env.render(mode='human_fast')
    print('Sim elapsed time:', laptime, 'Real elapsed time:', time.time() - start, 'steps:', steps,
          'laps:', obs['lap_counts'], 'collisions:', obs['collisions'])
    return steps, laptime, obs


if __name__ == '__main__':
    main()
