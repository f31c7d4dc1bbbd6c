"""One process, one GPU: a shard of B envs stepped as ONE batch against the same envs as TWO half-batches on two streams,
stepped alternately without joining (a policy that works on one half while the other half is being simulated):
    python tools/two_shards_probe.py [--envs 65536] [--steps 200]
The halves' launches overlap: one half's drain (the last ~100 us of a scan launch run at a third of the chip) is filled by
the other half's bulk.  Not the bench protocol (a step there is one pass over one batch) -- a usage note."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=65536)
ap.add_argument('--steps', type=int, default=200)
a = ap.parse_args()
B, K = a.envs, a.steps
poses = workload.spawn_poses(B, 1)
acts_np = workload.action_pool(8, B, 1)


def run(parts):
    n = B // parts
    envs, acts, streams = [], [], []
    for p in range(parts):
        e = F110VecEnv(n, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
        envs.append(e)
        acts.append(torch.as_tensor(acts_np[:, p * n:(p + 1) * n], device=e.device))
        streams.append(torch.cuda.Stream(device=e.device))
        with torch.cuda.stream(streams[p]):
            e.reset(torch.as_tensor(poses[p * n:(p + 1) * n], device=e.device))
    for k in range(60):
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                envs[p].step(acts[p][k % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                envs[p].step(acts[p][k % 8])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in envs:
        e.close()
    return B * K / dt / 1e6


for parts in (1, 2, 4, 1, 2):
    print('%d envs as %d batch(es) on %d stream(s): %.2f M env-steps/s' % (B, parts, parts, run(parts)), flush=True)
