"""Step-path timing on another bundled map (spawn = random free cells with clearance):
    python tools/sweep_map.py maps/berlin [envs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload  # noqa: E402
from red_gym_amd.maps import ASSETS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'maps/berlin'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env = F110VecEnv(B, map=os.path.join(ASSETS, name), num_agents=1, autoreset=True, count_lookups=True)
dt = env.eng.get_map_dt()
m = env.eng.map
rng = np.random.default_rng(0)
free = np.argwhere(dt > 0.5)
pick = free[rng.integers(0, len(free), B)]
poses = np.zeros((B, 1, 3))
poses[:, 0, 0] = m.orig_x + (pick[:, 1] + 0.5) * m.resolution
poses[:, 0, 1] = m.orig_y + (pick[:, 0] + 0.5) * m.resolution
poses[:, 0, 2] = rng.uniform(0, 6.28, B)
acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
env.reset(poses)
for k in range(40):
    env.step(acts[k % 8])
env.eng.t['lookups'].zero_()
env.eng.profile_begin(100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(100):
    env.step(acts[k % 8])
torch.cuda.synchronize()
dt_s = time.perf_counter() - t0
ms, n = env.eng.profile_end()
lk = env.eng.t['lookups'].to(torch.int64).sum().item() / (100 * B)
print('%-16s %dx%d res %.4f: ms/step %.3f scan_ms %.3f Msteps/s %.1f lookups/car-step %.0f' %
      (name, m.height, m.width, m.resolution, dt_s / 100 * 1e3, ms / n, B * 100 / dt_s / 1e6, lk), flush=True)
env.close()
