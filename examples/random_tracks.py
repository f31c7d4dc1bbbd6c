"""Domain randomisation on one GPU: K random tracks (the reference's unittest/random_trackgen.py walker, drawn and
turned into maps on the device), the envs split over them, EVERY env with its own vehicle (friction, cornering stiffness,
mass, inertia: the `params` a reference env is constructed with, f110_env.py:125-128) and one of 32 lidar-noise seeds
(:102-105), a GPU pure-pursuit policy per block, and the bird's-eye bitmap the reference's RL consumers build from every scan
(weap_util.lidar.lidar_to_bitmap) -- nothing leaves the GPU inside the loop.

    python examples/random_tracks.py [--envs 8192] [--tracks 8] [--steps 500]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, LidarBitmap, workload  # noqa: E402
from red_gym_amd.engine import DEFAULT_PARAMS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=8192)
    ap.add_argument('--tracks', type=int, default=8)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--seed', type=int, default=2025)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    vehicles = [dict(DEFAULT_PARAMS, mu=float(rng.uniform(0.8, 1.2)), C_Sf=float(rng.uniform(4.0, 5.4)), C_Sr=float(rng.uniform(4.6, 6.2)),
                     m=float(rng.uniform(3.3, 4.2)), I=float(rng.uniform(0.04, 0.055))) for _ in range(a.envs)]
    seeds = [a.seed + e % 32 for e in range(a.envs)]
    env = F110VecEnv(a.envs, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False, params=vehicles, seed=seeds)
    t0 = time.perf_counter()
    tracks, assign = env.randomize_tracks(range(a.seed, a.seed + a.tracks))
    torch.cuda.synchronize()
    print('%d tracks drawn and installed in %.1f ms' % (a.tracks, (time.perf_counter() - t0) * 1e3))
    poses = np.zeros((a.envs, 1, 3))
    for e in range(a.envs):
        wp = tracks[assign[e]].waypoints
        poses[e, 0] = wp[(e * 13) % len(wp)]
    racelines = [torch.as_tensor(np.column_stack([t.waypoints[:, :2], np.full(len(t.waypoints), 4.0)]), device=env.device)
                 for t in tracks]
    to_img = LidarBitmap(1080, bg_color='black', draw_mode='FILL', device=env.device)
    imgs = torch.empty((a.envs, 256, 256), dtype=torch.uint8, device=env.device)
    obs = env.reset(poses)[0]
    crashed = torch.zeros(a.envs, dtype=torch.bool, device=env.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        obs = env.step(env.pure_pursuit_blocks(racelines, assign, 1.5, 1.0))[0]
        to_img(obs['scans'][:, 0], out=imgs)
        crashed |= obs['collisions'][:, 0] > 0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d envs x %d steps on %d tracks: %.2f M env-steps/s incl. planner and bitmaps, %d cars touched a wall, '
          'mean drivable-area pixels %.0f' % (a.envs, a.steps, a.tracks, a.envs * a.steps / dt / 1e6, int(crashed.sum()),
                                                float((imgs > 0).float().sum(dim=(1, 2)).mean())))
    env.close()


if __name__ == '__main__':
    main()
