#!/bin/bash
# map per env: tests, then the cost of one-wave workgroups (two maps interleaved vs the same two maps in blocks)
set -e
mkdir -p gpurun_out/mapenv
timeout -k 10 500 python -m pytest tests/test_gpu_trackgen.py tests/test_gpu_fullsize.py -k "map or launch_order or 65536-1" -x -q > gpurun_out/mapenv/tests.log 2>&1
tail -3 gpurun_out/mapenv/tests.log
timeout -k 10 300 python - > gpurun_out/mapenv/ab.log 2>&1 <<'PY'
import time, numpy as np, torch
from red_gym_amd import F110VecEnv, workload, maps
B = 65536
env = F110VecEnv(B, map=workload.EXAMPLE_MAP, map_ext='.png', num_agents=1, autoreset=True)
m = maps.load_map(workload.EXAMPLE_MAP + '.yaml', '.png')
env.eng.set_map_occupancy(m.free, m.resolution, m.orig_x, m.orig_y, 0.0, slot=1)   # the same track twice: equal work per car
poses = torch.as_tensor(workload.spawn_poses(B, 1), device=env.device)
acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
def run(assign, label):
    env.eng.assign_maps(assign)
    env.reset(poses)
    for k in range(100): env.step(acts[k % 8])
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(400): env.step(acts[k % 8])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 400
    print('%-40s %.3f ms/step  %.1f M env-steps/s' % (label, dt * 1e3, B / dt / 1e6), flush=True)
run(None, 'one map')
run((np.arange(B) * 2) // B, 'two maps, blocks (2-wave workgroups)')
run(np.arange(B) % 2, 'two maps, interleaved (1-wave workgroups)')
run(None, 'one map (again)')
PY
cat gpurun_out/mapenv/ab.log
timeout -k 10 200 python bench.py > gpurun_out/mapenv/bench.json 2> gpurun_out/mapenv/bench.err
cat gpurun_out/mapenv/bench.json
