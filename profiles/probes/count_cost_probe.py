import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from red_gym_amd import F110VecEnv, workload
for B in (65536, 4096):
    for cnt in (False, True, False, True):
        env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True, count_lookups=cnt)
        env.reset(torch.as_tensor(workload.spawn_poses(B, 1), device=env.device))
        acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
        for k in range(60): env.step(acts[k % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(200): env.step(acts[k % 8])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(B, 'count_lookups', cnt, '%.4f ms/step  %.2f M/s' % (dt * 1e3, B / dt / 1e6), flush=True)
        env.close()
