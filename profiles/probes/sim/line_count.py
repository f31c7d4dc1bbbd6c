"""Offline estimate of the scan kernel's L1 work per gather under different cell-table layouts: replays the wave
scheduling policy (tools/sim/wave_schedule.py) with the actual cells every ray reads, and counts the distinct
64-B / 128-B lines each 64-lane gather touches.
    python tools/sim/line_count.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from red_gym_amd import workload  # noqa: E402
from red_gym_amd.maps import load_map  # noqa: E402
from wave_schedule import static_order  # noqa: E402


def beam_cells(dt, m, pose, nb=1080, fov=2 * np.pi, eps=1e-4, max_range=30.0):
    """per beam: list of (row, col) read while marching (after the first read at the car)"""
    H, W = dt.shape
    th = pose[2] - fov / 2 + np.arange(nb) * (fov / (nb - 1))
    c, s = np.cos(th), np.sin(th)
    ci0 = int(np.floor((pose[0] - m.orig_x) / m.resolution)); ri0 = int(np.floor((pose[1] - m.orig_y) / m.resolution))
    d0 = dt[ri0, ci0]
    x, y = pose[0] + d0 * c, pose[1] + d0 * s
    total = np.full(nb, d0)
    act = np.full(nb, d0 > eps and d0 <= max_range)
    cells = [[] for _ in range(nb)]
    while act.any():
        ci = np.floor((x - m.orig_x) / m.resolution).astype(np.int64)
        ri = np.floor((y - m.orig_y) / m.resolution).astype(np.int64)
        oob = (ci < 0) | (ci >= W) | (ri < 0) | (ri >= H)
        dd = np.where(oob, dt[-1, -1], dt[np.clip(ri, 0, H - 1), np.clip(ci, 0, W - 1)])
        for b in np.flatnonzero(act):
            cells[b].append((int(np.clip(ri[b], -1, H)), int(np.clip(ci[b], -1, W))))
        total = np.where(act, total + dd, total)
        x = np.where(act, x + dd * c, x)
        y = np.where(act, y + dd * s, y)
        act = act & (dd > eps) & (total <= max_range)
    return cells


LAYOUTS = {
    # name: (r, c) -> (64-B sector id, 128-B line id)
    'u16, 8-col strips (shipped): 128 B = 8x8 cells': lambda r, c: ((c >> 3, r >> 2), (c >> 3, r >> 3)),
    'u8, 8-col strips: 64 B = 8x8 cells, 128 B = 16x8': lambda r, c: ((c >> 3, r >> 3), (c >> 3, r >> 4)),
    'u8, 16-col strips: 64 B = 4x16, 128 B = 8x16': lambda r, c: ((c >> 4, r >> 2), (c >> 4, r >> 3)),
    'u16 row-major (round-1 first try)': lambda r, c: ((r, c >> 5), (r, c >> 6)),
}


def replay(cells, order, idle_min=40):
    """yields, per march iteration, the list of (r, c) read by the active lanes"""
    lane_beam = [-1] * 64
    lane_pos = [0] * 64
    nxt, nb = 0, len(order)
    while True:
        idle = [i for i in range(64) if lane_beam[i] < 0]
        if len(idle) >= idle_min or len(idle) == 64:
            for i in idle:
                while nxt < nb and not cells[order[nxt]]:
                    nxt += 1        # a beam that ends at its first read never marches
                if nxt >= nb:
                    break
                lane_beam[i], lane_pos[i] = order[nxt], 0
                nxt += 1
        if all(b < 0 for b in lane_beam):
            if nxt >= nb:
                return
            continue
        go = 64 - idle_min if nxt < nb else 0
        while True:
            reads = []
            for i in range(64):
                b = lane_beam[i]
                if b >= 0:
                    reads.append(cells[b][lane_pos[i]])
                    lane_pos[i] += 1
                    if lane_pos[i] >= len(cells[b]):
                        lane_beam[i] = -1
            yield reads
            if sum(b >= 0 for b in lane_beam) <= go:
                break


if __name__ == '__main__':
    from scipy.ndimage import distance_transform_edt
    m = load_map(workload.EXAMPLE_MAP + '.yaml', '.png')
    dt = m.resolution * distance_transform_edt(m.free)
    poses = workload.spawn_poses(24, 1)[:, 0]
    so = static_order()
    tot = {k: [0, 0] for k in LAYOUTS}
    gathers = lookups = 0
    for p in poses:
        cells = beam_cells(dt, m, p)
        for reads in replay(cells, so):
            gathers += 1
            lookups += len(reads)
            for name, f in LAYOUTS.items():
                ids = [f(r, c) for r, c in reads]
                tot[name][0] += len({i[0] for i in ids})
                tot[name][1] += len({i[1] for i in ids})
    print('%d cars: %.1f gathers per car, %.1f active lanes per gather' % (len(poses), gathers / len(poses), lookups / gathers))
    for name, (s64, l128) in tot.items():
        print('%-52s distinct 64-B sectors per gather %5.2f   128-B lines %5.2f' % (name, s64 / gathers, l128 / gathers))


def replay_quads(cells, order, idle_quads_min=10, group=4):
    """quad-granular policy: a group of `group` adjacent lanes takes `group` adjacent beams together, only when all of
    its lanes are idle; refill when >= idle_quads_min groups are idle."""
    ng = 64 // group
    lane_beam = [-1] * 64
    lane_pos = [0] * 64
    nxt, nb = 0, len(order)
    while True:
        idle_g = [g for g in range(ng) if all(lane_beam[g * group + j] < 0 for j in range(group))]
        if len(idle_g) >= idle_quads_min or len(idle_g) == ng:
            for g in idle_g:
                if nxt >= nb:
                    break
                for j in range(group):
                    if nxt < nb:
                        b = order[nxt]; nxt += 1
                        if cells[b]:
                            lane_beam[g * group + j], lane_pos[g * group + j] = b, 0
        if all(b < 0 for b in lane_beam):
            if nxt >= nb:
                return
            continue
        go_groups = ng - idle_quads_min if nxt < nb else 0
        while True:
            reads = []
            for i in range(64):
                b = lane_beam[i]
                if b >= 0:
                    reads.append((i, cells[b][lane_pos[i]]))
                    lane_pos[i] += 1
                    if lane_pos[i] >= len(cells[b]):
                        lane_beam[i] = -1
            yield reads
            busy_g = sum(any(lane_beam[g * group + j] >= 0 for j in range(group)) for g in range(ng))
            if busy_g <= go_groups:
                break


def replay_with_lanes(cells, order, idle_min=40):
    lane_beam = [-1] * 64
    lane_pos = [0] * 64
    nxt, nb = 0, len(order)
    while True:
        idle = [i for i in range(64) if lane_beam[i] < 0]
        if len(idle) >= idle_min or len(idle) == 64:
            for i in idle:
                while nxt < nb and not cells[order[nxt]]:
                    nxt += 1
                if nxt >= nb:
                    break
                lane_beam[i], lane_pos[i] = order[nxt], 0
                nxt += 1
        if all(b < 0 for b in lane_beam):
            if nxt >= nb:
                return
            continue
        go = 64 - idle_min if nxt < nb else 0
        while True:
            reads = []
            for i in range(64):
                b = lane_beam[i]
                if b >= 0:
                    reads.append((i, cells[b][lane_pos[i]]))
                    lane_pos[i] += 1
                    if lane_pos[i] >= len(cells[b]):
                        lane_beam[i] = -1
            yield reads
            if sum(b >= 0 for b in lane_beam) <= go:
                break


def quad_cost(reads, f, q=4):
    """sum over groups of q lanes of the distinct 128-B lines the group touches (the L1's per-quad work, if it
    processes a wave quad by quad)"""
    per = {}
    for lane, (r, c) in reads:
        per.setdefault(lane // q, set()).add(f(r, c)[1])
    return sum(len(v) for v in per.values())


if __name__ == '__main__':
    f16 = LAYOUTS['u16, 8-col strips (shipped): 128 B = 8x8 cells']
    f8 = LAYOUTS['u8, 8-col strips: 64 B = 8x8 cells, 128 B = 16x8']
    plain = np.arange(1080)
    for label, gen in [('lane-granular refill (shipped), static order', lambda cells: replay_with_lanes(cells, so)),
                       ('lane-granular refill, beam order 0..1079', lambda cells: replay_with_lanes(cells, plain)),
                       ('quad-granular (4 lanes take 4 adjacent beams), refill at 10 idle quads', lambda cells: replay_quads(cells, plain, 10, 4)),
                       ('quad-granular, refill at 6 idle quads', lambda cells: replay_quads(cells, plain, 6, 4)),
                       ('octet-granular (8 lanes), refill at 4 idle octets', lambda cells: replay_quads(cells, plain, 4, 8)),
                       ('pair-granular (2 lanes), refill at 16 idle pairs', lambda cells: replay_quads(cells, plain, 16, 2))]:
        g = lk = q16 = q8 = l16 = 0
        for p in poses[:12]:
            cells = beam_cells(dt, m, p)
            for reads in gen(cells):
                g += 1; lk += len(reads)
                q16 += quad_cost(reads, f16); q8 += quad_cost(reads, f8)
                l16 += len({f16(r, c)[1] for _, (r, c) in reads})
        n = 12
        print('%-72s gathers/car %6.1f  lanes/gather %4.1f  per-quad lines/car: u16 %6.0f  u8 %6.0f   distinct lines/car %6.0f' % (label, g / n, lk / g, q16 / n, q8 / n, l16 / n))


def replay_quads_lane_exit(cells, order, idle_min=40, group=4):
    """cheap variant: the march loop still leaves when >= idle_min LANES are idle (the kernel's existing scalar test);
    at the refill every idle lane finishes its beam, but only lanes of fully idle groups take new (adjacent) beams."""
    ng = 64 // group
    lane_beam = [-1] * 64
    lane_pos = [0] * 64
    nxt, nb = 0, len(order)
    while True:
        idle_g = [g for g in range(ng) if all(lane_beam[g * group + j] < 0 for j in range(group))]
        for g in idle_g:
            if nxt >= nb:
                break
            for j in range(group):
                if nxt < nb:
                    b = order[nxt]; nxt += 1
                    if cells[b]:
                        lane_beam[g * group + j], lane_pos[g * group + j] = b, 0
        if all(b < 0 for b in lane_beam):
            if nxt >= nb:
                return
            continue
        go = 64 - idle_min if nxt < nb else 0
        while True:
            reads = []
            for i in range(64):
                b = lane_beam[i]
                if b >= 0:
                    reads.append((i, cells[b][lane_pos[i]]))
                    lane_pos[i] += 1
                    if lane_pos[i] >= len(cells[b]):
                        lane_beam[i] = -1
            yield reads
            if sum(b >= 0 for b in lane_beam) <= go:
                break


if __name__ == '__main__':
    print()
    for label, gen in [('lane exit at 40 idle, take by fully idle quads', lambda cells: replay_quads_lane_exit(cells, plain, 40, 4)),
                       ('lane exit at 32 idle, take by fully idle quads', lambda cells: replay_quads_lane_exit(cells, plain, 32, 4)),
                       ('lane exit at 24 idle, take by fully idle quads', lambda cells: replay_quads_lane_exit(cells, plain, 24, 4)),
                       ('lane exit at 32 idle, take by fully idle pairs', lambda cells: replay_quads_lane_exit(cells, plain, 32, 2))]:
        g = lk = q16 = 0
        for p in poses[:12]:
            cells = beam_cells(dt, m, p)
            for reads in gen(cells):
                g += 1; lk += len(reads)
                q16 += quad_cost(reads, f16)
        print('%-72s gathers/car %6.1f  lanes/gather %4.1f  per-quad lines/car: u16 %6.0f' % (label, g / 12, lk / g, q16 / 12))


if __name__ == '__main__':
    # The counter calibrated (rocprofv3 on tools/ubench/gather_cost2): TCP_TOTAL_CACHE_ACCESSES of a 64-lane gather =
    # sum over its four 16-LANE groups of the distinct 128-B lines the group touches.  Same policies, that metric:
    print('\n16-lane-group metric (what TCP_TOTAL_CACHE_ACCESSES counts):')
    for label, gen in [('lane-granular taking, exit at 40 idle (shipped)', lambda cells: replay_with_lanes(cells, so)),
                       ('quad taking, exit at 32 idle lanes (measured: +8 % gathers, +3 % accesses)', lambda cells: replay_quads_lane_exit(cells, plain, 32, 4)),
                       ('8-lane taking, exit at 32 idle lanes', lambda cells: replay_quads_lane_exit(cells, plain, 32, 8)),
                       ('16-lane taking, exit at 32 idle lanes', lambda cells: replay_quads_lane_exit(cells, plain, 32, 16)),
                       ('16-lane taking, exit at 16 idle lanes', lambda cells: replay_quads_lane_exit(cells, plain, 16, 16)),
                       ('16-lane taking, refill when 1 group idle', lambda cells: replay_quads(cells, plain, 1, 16)),
                       ('16-lane taking, refill when 2 groups idle', lambda cells: replay_quads(cells, plain, 2, 16))]:
        g = lk = acc = 0
        for p in poses[:12]:
            cells = beam_cells(dt, m, p)
            for reads in gen(cells):
                g += 1; lk += len(reads)
                acc += quad_cost(reads, f16, 16)
        print('%-80s gathers/car %6.1f  lanes/gather %4.1f  L1 accesses/car %6.0f  per gather %5.1f' % (label, g / 12, lk / g, acc / 12, acc / g))


if __name__ == '__main__':
    print('\ncell layouts under the calibrated metric, shipped taking policy; footprint = distinct lines a car touches in one scan:')
    lay = {'u16, 8-col strips: 128 B = 8x8 cells (shipped)': lambda r, c: (c >> 3, r >> 3),
           'u8, 8-col strips: 128 B = 16 rows x 8 cols': lambda r, c: (c >> 3, r >> 4),
           'u8, 16-col strips: 128 B = 8 rows x 16 cols': lambda r, c: (c >> 4, r >> 3),
           'u8, 128 B = 12x10-ish (11 x 11 cells, 7 B wasted)': lambda r, c: (c // 11, r // 11),
           '4-bit codes: 128 B = 16 x 16 cells': lambda r, c: (c >> 4, r >> 4)}
    for name, f in lay.items():
        g = acc = foot = 0
        for p in poses[:12]:
            cells = beam_cells(dt, m, p)
            seen = set()
            for reads in replay_with_lanes(cells, so):
                g += 1
                per = {}
                for lane, (r, c) in reads:
                    per.setdefault(lane >> 4, set()).add(f(r, c))
                    seen.add(f(r, c))
                acc += sum(len(v) for v in per.values())
            foot += len(seen)
        print('%-52s L1 accesses/car %6.0f (per gather %5.2f)   lines touched per scan %5.0f' % (name, acc / 12, acc / g, foot / 12))
