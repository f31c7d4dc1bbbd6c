"""Per-step time of the first steps after construction + reset (is a short run slower because the GPU is cold, or
because the first steps do different work?):  python tools/step_ramp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from red_gym_amd import F110VecEnv, workload
B = 65536
env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True, count_lookups=True,
                 noise_std=float(os.environ.get('NOISE_STD', '0.01')))
dev = env.device
poses = torch.as_tensor(workload.spawn_poses(B, 1), device=dev)
acts = torch.as_tensor(workload.action_pool(16, B, 1), device=dev)
for rep in range(3):
    env.reset(poses)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
    lk = []
    torch.cuda.synchronize()
    ev[0].record()
    for k in range(120):
        env.eng.t['lookups'].zero_()
        env.step(acts[k % 16])
        ev[k + 1].record()
        lk.append(env.eng.t['lookups'].sum(dtype=torch.int64))
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(120)]
    lk = [int(v) / B for v in lk]
    print('rep', rep, 'ms/step by decade:', ' '.join('%.3f' % (sum(ms[i:i + 10]) / 10) for i in range(0, 120, 10)))
    print('rep', rep, 'lookups/car by decade:', ' '.join('%.0f' % (sum(lk[i:i + 10]) / 10) for i in range(0, 120, 10)))
    print('rep', rep, 'first 10 steps:', ' '.join('%.3f' % v for v in ms[:10]), flush=True)
    if rep == 0:
        time.sleep(2.0)  # let the GPU go idle again (rep 2 follows rep 1 without a pause)
