"""Classic step path against the car-group path (f110_set_step_path) in ONE process:
    python tools/group_sweep.py [sizes...] [--agents A] [--steps K]
For every size: ms/step of classic and of groups of 2 / 4 / 8 wavefronts per car, and a check that state, scans and
flags after the timed steps are IDENTICAL to the classic path's (same spawn poses, same action stream)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('sizes', nargs='*', type=int, default=[1024, 2048, 4096, 8192, 16384])
ap.add_argument('--agents', type=int, default=1)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--warmup', type=int, default=60)
ap.add_argument('--paths', default='classic,closed,group:2,group:4')
a = ap.parse_args()
for B in a.sizes:
    ref = None
    line = ['%6d x %d' % (B, a.agents)]
    for path in a.paths.split(','):
        name, _, w = path.partition(':')
        env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=a.agents, autoreset=True)
        env.eng.set_step_path(name, int(w or 0))
        poses = torch.as_tensor(workload.spawn_poses(B, a.agents), device=env.device)
        acts = torch.as_tensor(workload.action_pool(8, B, a.agents), device=env.device)
        env.reset(poses)
        for k in range(a.warmup):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        snap = {k: env.eng.t[k].clone() for k in ('state', 'scans', 'collisions', 'toggles', 'lap_times', 'done', 'noise_step',
                                                   'steer_buf', 'steer_cnt', 'current_time', 'pending_reset', 'pose_snap')}
        same = ''
        if ref is None:
            ref = snap
        else:
            bad = [k for k in snap if not torch.equal(snap[k], ref[k])]
            same = ' ==' if not bad else ' DIFFERS(%s)' % ','.join(bad)
        line.append('%s %.4f ms (%.1f M/s)%s' % (path, dt * 1e3, B / dt / 1e6, same))
        env.close()
    print('  '.join(line), flush=True)
