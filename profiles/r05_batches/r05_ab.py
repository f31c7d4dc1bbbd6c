import os, sys, time
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from red_gym_amd import F110VecEnv, workload
def run(B, A, reorder, steps=300, warm=100):
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=A, autoreset=True)
    env.eng.scan_reorder = reorder
    poses = torch.as_tensor(workload.spawn_poses(B, A), device=env.device)
    acts = torch.as_tensor(workload.action_pool(8, B, A), device=env.device)
    env.reset(poses)
    for k in range(warm): env.step(acts[k % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): env.step(acts[k % 8])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.close()
    return B * steps / dt / 1e6, dt / steps * 1e3
for r in range(2):
    for (B, A) in ((16384, 2), (65536, 1), (32768, 1)):
        for reorder in (False, True):
            m, ms = run(B, A, reorder)
            print('%6d x %d  reorder %-5s  %.2f M env-steps/s  %.4f ms/step' % (B, A, reorder, m, ms), flush=True)
