"""Times the step path for one build variant of the library (F110_LIB env var):
    F110_LIB=build_variants/x.so python tools/sweep.py [--envs B] [--agents A] [--steps K]
Prints one line: variant, ms/step, scan-kernel ms."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=65536)
ap.add_argument('--agents', type=int, default=1)
ap.add_argument('--steps', type=int, default=60)
ap.add_argument('--warmup', type=int, default=40)
a = ap.parse_args()
env = F110VecEnv(a.envs, map=workload.EXAMPLE_MAP, num_agents=a.agents, autoreset=True, count_lookups=True)
dev = env.device
poses = torch.as_tensor(workload.spawn_poses(a.envs, a.agents), device=dev)
acts = torch.as_tensor(workload.action_pool(8, a.envs, a.agents), device=dev)
env.reset(poses)
for k in range(a.warmup):
    env.step(acts[k % 8])
env.eng.t['lookups'].zero_()
env.eng.profile_begin(a.steps)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(a.steps):
    env.step(acts[k % 8])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = env.eng.profile_end()
cnt = env.eng.t['lookups'].to(torch.int64).sum().item() / (a.steps * a.envs * a.agents)
print('%-40s ms/step %.3f  scan_ms %.3f  Msteps/s %.2f  count/car-step %.1f' % (os.path.basename(os.environ.get('F110_LIB', 'default')),
      dt / a.steps * 1e3, ms / n, a.envs * a.steps / dt / 1e6, cnt), flush=True)
env.close()
