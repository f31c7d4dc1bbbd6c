"""Runs the function-level scan (f110_scan) on the benchmark spawn poses a few times
(for rocprofv3 --pmc passes that isolate the march's L1 behaviour)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from red_gym_amd import workload  # noqa: E402
from red_gym_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
e = Engine(num_envs=1, num_agents=1, noise_std=0)
e.set_map(workload.EXAMPLE_MAP + '.yaml', '.png')
poses = torch.as_tensor(workload.spawn_poses(n, 1)[:, 0], device=e.device)
out = None
for k in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, out32 = e.scan(poses, want_f32=True)
    torch.cuda.synchronize()
    print('scan %d poses: %.3f ms' % (n, (time.perf_counter() - t0) * 1e3), flush=True)
e.close()
