import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from red_gym_amd import F110VecEnv, workload
for B in (65536, 4096):
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
    env.reset(torch.as_tensor(workload.spawn_poses(B, 1), device=env.device))
    acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
    for k in range(30): env.step(acts[k % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200): env.step(acts[k % 8])
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    buf = env.capture_step()
    for k in range(10): env.step_graph(acts[k % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200): env.step_graph(acts[k % 8])
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 200
    print(B, 'eager %.4f ms  graph %.4f ms' % (eager * 1e3, graph * 1e3))
    env.close()
