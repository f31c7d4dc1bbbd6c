"""SURVEY H3(c) probe: does it help the scan kernel's L1 when the cars of a launch are ordered by map region?
    python tools/region_order_probe.py [--envs 65536] [--sort 0|1] [--region 64] [--every 8] [--steps 48]
With --sort 1 the envs are physically re-ordered every `--every` steps by the 64x64-cell region of their pose (all
per-env tensors are permuted together; envs are independent, so this IS launching the cars in region order, without
touching the kernel): neighbouring waves -- the 8 waves of a SIMD, the 32 of a CU -- then march rays through the same
part of the table.  Prints the mean scan_kernel time by events; run it under `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum` for the counters (the permutation itself runs in torch kernels, which the summary ignores)."""
import argparse
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=65536)
ap.add_argument('--sort', type=int, default=1)
ap.add_argument('--region', type=int, default=64, help='region edge in cells')
ap.add_argument('--every', type=int, default=8)
ap.add_argument('--steps', type=int, default=48)
ap.add_argument('--warmup', type=int, default=40)
a = ap.parse_args()
B = a.envs
env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True, count_lookups=True)
dev = env.device
poses = torch.as_tensor(workload.spawn_poses(B, 1), device=dev)
acts = torch.as_tensor(workload.action_pool(8, B, 1), device=dev)
env.reset(poses)
m = env.eng.map
res, ox, oy, W = m.resolution, m.orig_x, m.orig_y, m.width


def reorder():
    st = env.eng.t['state'][:, 0]
    cx = ((st[:, 0] - ox) / res).floor().long().clamp(0, W - 1) // a.region
    cy = ((st[:, 1] - oy) / res).floor().long().clamp(0, m.height - 1) // a.region
    # heading octant as the minor key: cars of a region looking the same way march the same cells
    octant = ((st[:, 4] % (2 * torch.pi)) / (2 * torch.pi) * 8).floor().long().clamp(0, 7)
    key = (cy * (W // a.region + 1) + cx) * 8 + octant
    perm = torch.argsort(key, stable=True)
    for k, t in env.eng.t.items():
        if t is not None and t.shape[0] == B:
            t.copy_(t.index_select(0, perm))
    return int(torch.unique(key).numel())


for k in range(a.warmup):
    env.step(acts[k % 8])
nreg = reorder() if a.sort else 0
env.eng.t['lookups'].zero_()
env.eng.profile_begin(a.steps)
for k in range(a.steps):
    if a.sort and k and k % a.every == 0:
        reorder()
    env.step(acts[k % 8])   # (the action rows stay with their slot, not with their env: random actions either way)
torch.cuda.synchronize()
ms, n = env.eng.profile_end()
lk = env.eng.t['lookups'].to(torch.int64).sum().item() / (a.steps * B)
print('envs %d  sort %d (region %d cells, every %d steps, %d distinct keys)  scan_kernel %.4f ms per launch over %d launches, %.1f lookups per car'
      % (B, a.sort, a.region, a.every, nreg, ms / n, n, lk), flush=True)
env.close()
