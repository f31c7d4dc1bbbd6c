"""Per-wave timeline of ONE step's scan launch (diagnostics build of the library):
    hipcc ... -DF110_TIMELINE -o build_variants/timeline.so ; F110_LIB=build_variants/timeline.so python tools/timeline.py
    [--envs B] [--stages SPEC]
Prints, in microseconds from the first wave's start: when waves start (percentiles), wave lifetimes by kind (whole
car / part of a car), the end of the launch, and how many waves are resident over time."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from red_gym_amd import F110VecEnv, _lib, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--stages', default='')
ap.add_argument('--warm', type=int, default=60)
ap.add_argument('--agents', type=int, default=1)
a = ap.parse_args()
lib = _lib.load()
lib.f110_debug_timeline.argtypes = [C.c_void_p, C.c_int64]
B = a.envs
env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=a.agents, autoreset=True, count_lookups=True)
name = 'classic'
if a.stages:
    env.eng.set_scan_stages(a.stages)
env.reset(torch.as_tensor(workload.spawn_poses(B, a.agents), device=env.device))
acts = torch.as_tensor(workload.action_pool(8, B, a.agents), device=env.device)
for k in range(a.warm):
    env.step(acts[k % 8])
torch.cuda.synchronize()
env.eng.t['lookups'].zero_()
env.step(acts[7])
torch.cuda.synchronize()
prev_lookups = env.eng.t['lookups'].clone().reshape(-1).cpu().numpy().astype(np.int64)  # the step before the stamped one
env.eng.t['lookups'].zero_()
env.step(acts[0])
torch.cuda.synchronize()
lookups = env.eng.t['lookups'].reshape(-1).cpu().numpy().astype(np.int64)
N = 1 << 18
buf = np.zeros((N, 4), dtype=np.uint64)
assert lib.f110_debug_timeline(buf.ctypes.data_as(C.c_void_p), N) == 0
buf = buf[buf[:, 0] != 0]
# the top 16 bits of the three stamps carry counts: march iterations, passes of the outer loop, refill phases (beams taken)
M48 = np.uint64((1 << 48) - 1)
march_it = (buf[:, 0] >> np.uint64(48)).astype(np.int64)
phases = (buf[:, 1] >> np.uint64(48)).astype(np.int64)
refills = (buf[:, 2] >> np.uint64(48)).astype(np.int64)
buf[:, 0] &= M48; buf[:, 1] &= M48; buf[:, 2] &= M48
t0 = buf[:, 0].min()
start = (buf[:, 0] - t0).astype(np.float64) / 100.0   # 100 MHz ticks -> us
ready = (buf[:, 1] - t0).astype(np.float64) / 100.0
end = (buf[:, 2] - t0).astype(np.float64) / 100.0
# car << 8 | part | wpc << 40 | march iterations after the wave's queue ran dry << 44
wpc = ((buf[:, 3] >> np.uint64(40)) & np.uint64(0xf)).astype(np.int64)
car = ((buf[:, 3] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
drain_it = (buf[:, 3] >> np.uint64(44)).astype(np.int64)
ncars = len(set(car.tolist()))
print('  per car (sum over its waves): march iterations %.1f, refill phases %.1f, passes of the outer loop %.1f; per wave: %.1f / %.1f / %.1f; %d cars, %d waves'
      % (march_it.sum() / ncars, refills.sum() / ncars, phases.sum() / ncars, march_it.mean(), refills.mean(), phases.mean(), ncars, len(buf)))
print('%s envs, lib %s, stages=%r: %d waves, launch ends at %.1f us after the first wave starts' % (B, os.path.basename(os.environ.get('F110_LIB', 'default')), a.stages, len(buf), end.max()))
pc = lambda x: ' '.join('%.1f' % v for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))  # noqa: E731
print('  wave start      p0/10/50/90/99/100: ' + pc(start))
print('  start -> queue dry (whole life of a wave whose car has no ray to march): ' + pc(ready - start))
for k in sorted(set(wpc.tolist())):
    m = wpc == k
    print('  %d wave(s)/car: %6d waves, lifetime %s   end %s' % (k, m.sum(), pc((end - start)[m]), pc(end[m])))
# per car: end of its last wave
last = {}
for c, e in zip(car.tolist(), end.tolist()):
    last[c] = max(last.get(c, 0.0), e)
ce = np.array(list(last.values()))
print('  car completion  p0/10/50/90/99/100: ' + pc(ce))
# the drain phase of a wave (queue dry -> last ray done): microseconds per march iteration = the dependent chain of ONE
# table lookup, for all waves and for the waves that end last (nearly alone on the chip)
drain_us = end - ready
ok = drain_it > 20
if ok.sum():
    print('  march iterations after the queue ran dry: ' + pc(drain_it[ok]))
    print('  us per iteration in the drain phase, p10/50/90 over waves: ' + ' '.join('%.3f' % v for v in np.percentile((drain_us / np.maximum(drain_it, 1))[ok], [10, 50, 90])))
    lastw = np.argsort(end)[-12:]
    print('  the 12 waves that end last: ' + '  '.join('end %.0f us: drain %d it @ %.3f us' % (end[i], drain_it[i], drain_us[i] / max(drain_it[i], 1)) for i in lastw))
# the longest-living waves: does the car's lookup count (this step / the step before) predict them?
life = end - start
top = np.argsort(life)[-12:][::-1]
rank_now = np.argsort(np.argsort(-lookups)); rank_prev = np.argsort(np.argsort(-prev_lookups))
print('  longest-living waves: ' + '  '.join('%.0f us (start %.0f, %d w/car) car lookups %d = rank %d now, %d before' % (
    life[i], start[i], wpc[i], lookups[car[i]], rank_now[car[i]], rank_prev[car[i]]) for i in top))
w1 = wpc == 1
if w1.sum() > 100:
    cc = np.corrcoef(life[w1], lookups[car[w1]])[0, 1]
    cp = np.corrcoef(life[w1], prev_lookups[car[w1]])[0, 1]
    print('  whole-car waves: correlation of lifetime with the car\'s lookups %.3f (this step), %.3f (previous step); lookups per car p50/p99/max %d %d %d'
          % (cc, cp, np.percentile(lookups, 50), np.percentile(lookups, 99), lookups.max()))
# workgroups of the classic launch hold SCAN_WAVES = 2 consecutive waves: a slot whose wave is done waits for the workgroup's other
# wave before the workgroup's LDS is free for the next one
if os.environ.get('F110_SCAN_WAVES', '2') == '2':
    part = (buf[:, 3] & np.uint64(0xff)).astype(np.int64)
    key = np.where(wpc == 1, car // 2, -1 - (car * 8 + part // 2))  # whole cars: cars 2k, 2k+1; split cars: parts 2j, 2j+1 of a car
    o = np.argsort(key, kind='stable')
    ks, es, ss = key[o], end[o], start[o]
    same = ks[1:] == ks[:-1]
    i0 = np.nonzero(same)[0]
    spread = np.abs(es[i0] - es[i0 + 1])
    life2 = np.maximum(es[i0], es[i0 + 1]) - np.minimum(ss[i0], ss[i0 + 1])
    print('  workgroups (%d pairs of waves): |end - end| p10/50/90 %s us; a slot waits for its workgroup\'s other wave %.1f %% of the pairs\' slot-time'
          % (len(i0), ' '.join('%.1f' % v for v in np.percentile(spread, [10, 50, 90])), 100.0 * spread.sum() / (2.0 * life2.sum())))
grid = np.arange(0.0, end.max() + 5.0, 5.0)
res = [(int(((start <= t) & (end > t)).sum())) for t in grid]
print('  resident waves every 5 us (mean over the middle half of the launch %.0f): ' % np.mean(res[len(res) // 4: 3 * len(res) // 4]) + ' '.join(str(r) for r in res))
env.close()
