"""ms/step of the classic path for a list of scan stage lists (f110_set_scan_stages), one process:
    python tools/stage_sweep.py --envs 4096 '*:0' '*:2' '2048:0,*:2' ..."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('specs', nargs='+')
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--steps', type=int, default=200)
a = ap.parse_args()
B = a.envs
env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
poses = torch.as_tensor(workload.spawn_poses(B, 1), device=env.device)
acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
for rep in range(2):
    for spec in a.specs:
        env.eng.set_scan_stages(None if spec == 'default' else spec)
        env.reset(poses)
        for k in range(60):
            env.step(acts[k % 8])
        env.eng.profile_begin(a.steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        ms, n = env.eng.profile_end()
        print('%6d  %-24s step %.4f ms  scan %.4f ms  (%.1f M/s)' % (B, spec, dt * 1e3, ms / n, B / dt / 1e6), flush=True)
env.close()
