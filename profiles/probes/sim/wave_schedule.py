"""Offline model of scan_kernel's lane scheduling: how many march iterations (= wave-wide gathers) a car costs
under different beam orders / refill thresholds, from exact per-beam lookup counts computed on the CPU.
    python tools/sim/wave_schedule.py
Per-beam counts come from a NumPy restatement of the march (laser_models.py:107-146) on the reference's distance
table; the model replays scan_kernel's policy: 64 lanes, refill when >= `idle_min` lanes are idle (or no beam is
marching), beams taken in the given order, drain once the order is exhausted."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from red_gym_amd import workload  # noqa: E402
from red_gym_amd.maps import load_map  # noqa: E402


def beam_counts(dt, m, pose, nb=1080, fov=2 * np.pi, eps=1e-4, max_range=30.0):
    """march steps per beam AFTER the first table read (vectorised over beams)"""
    H, W = dt.shape
    th = pose[2] - fov / 2 + np.arange(nb) * (fov / (nb - 1))
    c, s = np.cos(th), np.sin(th)

    def look(x, y):
        ci = np.floor((x - m.orig_x) / m.resolution).astype(np.int64)
        ri = np.floor((y - m.orig_y) / m.resolution).astype(np.int64)
        oob = (ci < 0) | (ci >= W) | (ri < 0) | (ri >= H)
        return np.where(oob, dt[-1, -1], dt[np.clip(ri, 0, H - 1), np.clip(ci, 0, W - 1)])
    d0 = look(np.array([pose[0]]), np.array([pose[1]]))[0]
    x, y = pose[0] + d0 * c, pose[1] + d0 * s
    total = np.full(nb, d0)
    d = np.full(nb, d0)
    n = np.zeros(nb, dtype=np.int64)
    act = (d > eps) & (total <= max_range)
    while act.any():
        dd = look(x, y)
        n += act
        d = np.where(act, dd, d)
        total = np.where(act, total + dd, total)
        x = np.where(act, x + dd * c, x)
        y = np.where(act, y + dd * s, y)
        act = act & (dd > eps) & (total <= max_range)
    return n


def iterations(L, order, idle_min=40):
    """wave iterations for beams with march lengths L taken in `order`"""
    rem = np.zeros(64, dtype=np.int64)
    nxt, it, drain, nb = 0, 0, 0, len(order)
    while True:
        idle = rem == 0
        if idle.sum() >= idle_min or not (~idle).any():
            k = min(int(idle.sum()), nb - nxt)
            slots = np.flatnonzero(idle)[:k]
            rem[slots] = L[order[nxt:nxt + k]]
            nxt += k
        if not (rem > 0).any():
            if nxt >= nb:
                return it, drain
            continue
        go = 64 - idle_min if nxt < nb else 0
        while True:
            rem = np.maximum(rem - 1, 0)
            it += 1
            drain += nxt >= nb
            if (rem > 0).sum() <= go:
                break


def static_order(nb=1080, fov=2 * np.pi):
    nfull = nb // 64
    key = sorted((abs(np.sin(-fov / 2 + (64 * c + 31.5) * fov / (nb - 1))), 64 * c) for c in range(nfull))
    o = np.concatenate([np.arange(b, b + 64) for _, b in key] + [np.arange(64 * nfull, nb)])
    return o


if __name__ == '__main__':
    from scipy.ndimage import distance_transform_edt
    m = load_map(workload.EXAMPLE_MAP + '.yaml', '.png')
    dt = m.resolution * distance_transform_edt(m.free)
    poses = workload.spawn_poses(96, 1)[:, 0]
    Ls = [beam_counts(dt, m, p) for p in poses]
    prev = [beam_counts(dt, m, p + np.array([0.05, 0.03, 0.01])) for p in poses]  # "previous step": 6 cm, 0.6 degrees away
    so = static_order()
    rows = []
    for name, f in [('static |sin| chunk order (shipped)', lambda L, P: so),
                    ('beam order 0..1079', lambda L, P: np.arange(1080)),
                    ('chunks sorted by TRUE chunk total (oracle)', lambda L, P: np.concatenate([np.arange(64 * c, min(64 * c + 64, 1080)) for c in np.argsort(-np.array([L[64 * c:64 * c + 64].sum() for c in range(17)]))])),
                    ('chunks sorted by PREVIOUS-step chunk max', lambda L, P: np.concatenate([np.arange(64 * c, min(64 * c + 64, 1080)) for c in np.argsort(-np.array([P[64 * c:64 * c + 64].max() for c in range(17)]))])),
                    ('beams sorted by PREVIOUS-step length', lambda L, P: np.argsort(-P, kind='stable')),
                    ('beams sorted by TRUE length (LPT bound)', lambda L, P: np.argsort(-L, kind='stable')),
                    ('two buckets by previous length (> 12 first)', lambda L, P: np.concatenate([np.flatnonzero(P > 12), np.flatnonzero(P <= 12)]))]:
        its = np.array([iterations(L, f(L, P)) for L, P in zip(Ls, prev)])
        rows.append((name, its[:, 0].mean(), its[:, 1].mean()))
    ideal = np.mean([L.sum() / 64 for L in Ls])
    print('march steps per car (mean): %.0f   perfect packing: %.1f iterations' % (np.mean([L.sum() for L in Ls]), ideal))
    for name, a, b in rows:
        print('%-48s iterations %6.1f  of which drain %5.1f' % (name, a, b))
    for idle_min in (16, 24, 32, 40, 48):
        its = np.array([iterations(L, so, idle_min) for L in Ls])
        print('static order, refill at >= %2d idle lanes: iterations %6.1f  drain %5.1f  refill-phase bound %5.1f' % (idle_min, its[:, 0].mean(), its[:, 1].mean(), 1080 / idle_min))
    # --- chunk orders from what the GPU can know cheaply about the previous step ---
    def chunk_order(keys):
        return np.concatenate([np.arange(64 * c, min(64 * c + 64, 1080)) for c in np.argsort(-np.asarray(keys), kind='stable')])
    for idle_min in (24, 32, 40):
        its = np.array([iterations(L, chunk_order([P[64 * c:64 * c + 64].max() for c in range(17)]), idle_min) for L, P in zip(Ls, prev)])
        print('previous-step chunk max, refill at >= %2d idle: iterations %6.1f  drain %5.1f' % (idle_min, its[:, 0].mean(), its[:, 1].mean()))
    its = np.array([iterations(L, chunk_order([P[64 * c:64 * c + 64].sum() for c in range(17)])) for L, P in zip(Ls, prev)])
    print('previous-step chunk SUM:                        iterations %6.1f  drain %5.1f' % (its[:, 0].mean(), its[:, 1].mean()))
    its = np.array([iterations(L, chunk_order([np.sort(P[64 * c:64 * c + 64])[-4:].mean() for c in range(17)])) for L, P in zip(Ls, prev)])
    print('previous-step chunk mean of top 4:              iterations %6.1f  drain %5.1f' % (its[:, 0].mean(), its[:, 1].mean()))
    # a stale key (car was reset / moved a lot): keys from a pose 1 m and 0.5 rad away
    far = [beam_counts(dt, m, p + np.array([0.7, 0.7, 0.5])) for p in poses[:48]]
    its = np.array([iterations(L, chunk_order([P[64 * c:64 * c + 64].max() for c in range(17)])) for L, P in zip(Ls[:48], far)])
    print('keys from a pose 1 m / 0.5 rad away:            iterations %6.1f  drain %5.1f' % (its[:, 0].mean(), its[:, 1].mean()))
    # half-chunks (32 beams) as the sorting unit
    def half_order(P):
        keys = [P[32 * c:32 * c + 32].max() for c in range(34)]
        return np.concatenate([np.arange(32 * c, min(32 * c + 32, 1080)) for c in np.argsort(-np.asarray(keys), kind='stable')])
    its = np.array([iterations(L, half_order(P)) for L, P in zip(Ls, prev)])
    print('previous-step max of 32-beam half chunks:       iterations %6.1f  drain %5.1f' % (its[:, 0].mean(), its[:, 1].mean()))
