"""CPU model of the scan kernel's lane scheduling, to count L1 (TCP) accesses per gather
under different cell-table layouts and refill policies before touching the kernel.
Cost model from tools/ubench/gather_cost.hip (patterns 0-14): a 64-lane gather is priced
per QUAD of lanes -- one access per distinct 128-B line among a quad's active lanes
(quads only merge further when their addresses are contiguous in lane order, which a
ray march never produces)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from red_gym_amd import workload  # noqa: E402

sc = oracle.Scanner(1080, 2 * np.pi)
sc.set_map(workload.EXAMPLE_MAP + '.yaml', '.png')
m = sc.map
dt, res, ox, oy = m['dt'], m['resolution'], m['orig_x'], m['orig_y']
H, W = dt.shape


def trace_cells(pose):
    """per beam: list of (r, c) cells looked up AFTER the first (shared) lookup."""
    idx = sc.beam_indices(pose)
    out = []
    x0, y0 = pose[0], pose[1]
    r0, c0 = int((y0 - oy) / res), int((x0 - ox) / res)
    d0 = dt[r0, c0]
    for b in range(1080):
        c, s = sc.cosines[idx[b]], sc.sines[idx[b]]
        x, y, d, tot = x0, y0, d0, d0
        cells = []
        while d > 1e-4 and tot <= 30.0:
            x += d * c
            y += d * s
            xr, yr = x - ox, y - oy
            if xr < 0 or xr >= W * res or yr < 0 or yr >= H * res:
                r_, c_ = -1, -1
                d = dt[-1, -1]
            else:
                r_, c_ = int(yr / res), int(xr / res)
                d = dt[r_, c_]
            tot += d
            cells.append((r_, c_))
        out.append(cells)
    return out


def line_id(r, c, layout):
    th, tw = layout
    return (r // th) * 100000 + (c // tw)


def simulate(cells, layout, refill_idle=40, chunk=64, order='sin', quad_refill=False):
    nb = len(cells)
    angles = -np.pi + np.arange(nb) * (2 * np.pi / (nb - 1))
    nfull = nb // chunk
    keys = sorted(range(nfull), key=lambda k: abs(np.sin(angles[min(nb - 1, int(chunk * k + chunk / 2))])))
    chunks = keys + ([nfull] if nb % chunk else [])
    queue = [b for k in chunks for b in range(k * chunk, min(nb, (k + 1) * chunk))] if order == 'sin' else list(range(nb))
    lane_beam = [-1] * 64
    lane_pos = [0] * 64
    nxt = 0
    accesses = gathers = iters = lane_iters = 0
    while True:
        idle = [l for l in range(64) if lane_beam[l] < 0]
        if quad_refill:  # only quads whose four lanes are all idle take (four consecutive) beams
            idle = [l for l in idle if all(lane_beam[4 * (l // 4) + j] < 0 for j in range(4))]
        for l in idle:
            if nxt < nb:
                b = queue[nxt]; nxt += 1
                if len(cells[b]) > 0:
                    lane_beam[l], lane_pos[l] = b, 0
                # zero-length beams finish immediately
        if all(b < 0 for b in lane_beam):
            if nxt >= nb:
                break
            continue
        go = 64 - refill_idle if nxt < nb else 0
        while True:
            act = [l for l in range(64) if lane_beam[l] >= 0]
            if quad_refill and nxt < nb:
                free_q = sum(1 for q in range(16) if all(lane_beam[4 * q + j] < 0 for j in range(4)))
                if free_q * 4 >= refill_idle or not act:
                    break
            elif len(act) <= go or not act:
                break
            iters += 1
            gathers += 1
            lane_iters += len(act)
            for g in range(16):
                lines = set()
                for l in act:
                    if l // 4 == g:
                        r, c = cells[lane_beam[l]][lane_pos[l]]
                        if r >= 0:
                            lines.add(line_id(r, c, layout))
                accesses += len(lines)
            for l in act:
                lane_pos[l] += 1
                if lane_pos[l] >= len(cells[lane_beam[l]]):
                    lane_beam[l] = -1
    return accesses, gathers, lane_iters


if __name__ == '__main__':
    poses = workload.spawn_poses(24, 1)[:, 0]
    allc = [trace_cells(p) for p in poses]
    print('mean march lookups per car', np.mean([sum(len(c) for c in cl) for cl in allc]))
    for name, layout in [('u16 8x8', (8, 8)), ('u8 8x16', (8, 16))]:
        for quad in (False, True):
            for refill in (8, 16, 24, 40):
                tot = np.array([simulate(cl, layout, refill, quad_refill=quad) for cl in allc]).sum(axis=0)
                print('%-10s quad_refill=%d refill>=%2d: accesses/gather %5.1f  gathers/car %6.1f  accesses/car %7.0f  lane-util %.2f'
                      % (name, quad, refill, tot[0] / tot[1], tot[1] / len(allc), tot[0] / len(allc), tot[2] / tot[1] / 64))
