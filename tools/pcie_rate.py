"""What the step costs when actions come from and observations go back to HOST memory every step (the boundary itself
hands over device buffers; this is the PCIe-inclusive figure DESIGN.md section 6 quotes):  python tools/pcie_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from red_gym_amd import F110VecEnv, workload
B = 65536
env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
env.reset(torch.as_tensor(workload.spawn_poses(B, 1), device=env.device))
acts = torch.as_tensor(workload.action_pool(8, B, 1)).pin_memory()
scans_h = torch.empty((B, 1, 1080), dtype=torch.float32).pin_memory()
state_h = torch.empty((B, 1, 7), dtype=torch.float64).pin_memory()
for k in range(40):
    env.step(acts[k % 8].to(env.device, non_blocking=True))
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 30
for k in range(N):
    obs, _, done, _ = env.step(acts[k % 8].to(env.device, non_blocking=True))
    scans_h.copy_(obs['scans'], non_blocking=True)
    state_h.copy_(env.state, non_blocking=True)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print('65536 envs, actions H2D (1 MB) + fp32 scans D2H (283 MB) + state D2H (3.7 MB) every step, pinned host buffers: '
      '%.2f ms/step = %.1f M env-steps/s (%.1f GB/s over PCIe)' % (dt * 1e3, B / dt / 1e6, (283.1 + 3.7 + 1.0) / 1e3 / dt))
env.close()
