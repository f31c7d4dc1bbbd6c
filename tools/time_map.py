"""Times map installation (mask -> EDT -> cell codes, LUT, fp64 table) for the shipped maps:
    python tools/time_map.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, workload, maps  # noqa: E402

env = F110VecEnv(16, map=workload.EXAMPLE_MAP, num_agents=1)
for name in ['example_map', 'berlin', 'skirk', 'vegas']:
    y = workload.EXAMPLE_MAP + '.yaml' if name == 'example_map' else maps.builtin_map_yaml(name)
    m = maps.load_map(y, '.png')
    th = 0.0
    free_dev = torch.as_tensor(m.free, device='cuda')
    for label, arg in (('host mask', m.free), ('device mask', free_dev)):
        env.update_map_occupancy(arg, m.resolution, m.orig_x, m.orig_y, th)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            env.update_map_occupancy(arg, m.resolution, m.orig_x, m.orig_y, th)
        torch.cuda.synchronize()
        print('%-12s %4dx%-4d %-11s %.2f ms per install' % (name, m.height, m.width, label, (time.perf_counter() - t0) / 5 * 1e3))

for seed in (123, 7):
    env.randomize_track(seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5):
        env.randomize_track(seed + k)
    torch.cuda.synchronize()
    print('random track (centre line on the host, walls + EDT + tables on the GPU): %.2f ms per track'
          % ((time.perf_counter() - t0) / 5 * 1e3))
