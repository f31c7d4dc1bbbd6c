"""BASELINE.json's full sizes (4 096 and 65 536 envs x 1 agent, 16 384 envs x 2 agents) and the largest single-GPU shapes
(524 288 envs x 1 agent -- config 5's total on one GPU, the size at which the scan switches to its plain-store instantiation
by itself -- and 32 768 envs x 8 agents = 262 144 cars, 1.8 M car-opponent pairs): the oracle
cannot step them in seconds, so parity is checked on a seeded SAMPLE of envs against
independent oracle envs, and on size-independent properties of the whole batch:
determinism (two engines, same inputs -> identical bits), env independence (permuting
the envs permutes the outputs), and autoreset bookkeeping consistency."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def _mk(B, A, **kw):
    from red_gym_amd import F110VecEnv, workload
    return F110VecEnv(B, map=workload.EXAMPLE_MAP, map_ext='.png', num_agents=A, **kw)


@pytest.mark.parametrize('B,A', [(4096, 1), (65536, 1), (16384, 2), (524288, 1), (32768, 8)])
def test_fullsize_sample_vs_oracle_and_lookup_count(B, A):
    from red_gym_amd import workload
    T = 6 if B * A <= 65536 else 4
    env = _mk(B, A, autoreset=True, keep_f64_scans=True, count_lookups=True)
    poses = workload.spawn_poses(B, A)
    acts = workload.action_pool(T, B, A)
    rng = np.random.default_rng(9)
    # random envs plus the ends of the launch: the last min(2 048, N/2) cars run as four short waves each
    # (launch_scan; at 4 096 x 1 that is cars 2 048.., i.e. 2 048 whole-car waves + 2 048 cars split x4), so the
    # sample straddles that boundary on purpose
    tail0 = B - min(2048, B * A // 2) // A
    sample = np.unique(np.r_[rng.choice(B, size=40 if A < 4 else 10, replace=False), 0, 1, tail0 - 2, tail0 - 1, tail0, tail0 + 1, B - 2, B - 1])
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(workload.EXAMPLE_MAP + '.yaml', '.png')
    noise = oracle.noise_table(12345, T + 2)
    ors = {int(b): oracle.Env(sc, A, noise=noise) for b in sample}
    env.reset(torch.as_tensor(poses, device=env.device))
    oo = {b: ors[b].reset(poses[b]) for b in ors}
    pending = {b: oo[b]['done'] for b in ors}
    for k in range(T):
        obs, _, done, info = env.step(torch.as_tensor(acts[k], device=env.device))
        st, s64, s32 = _np(env.state[sample]), _np(obs['scans_f64'][sample]), _np(obs['scans'][sample])
        col, dn = _np(obs['collisions'][sample]), _np(done[sample])
        for j, b in enumerate(ors):
            oo[b] = ors[b].reset(poses[b]) if pending[b] else ors[b].step(acts[k][b])
            pending[b] = oo[b]['done']
            assert np.allclose(st[j], oo[b]['state'], rtol=0, atol=1e-9), (k, b)
            assert np.allclose(s64[j], oo[b]['scans'], rtol=0, atol=1e-9), (k, b)
            assert np.allclose(s32[j], oo[b]['scans'], rtol=0, atol=1e-5), (k, b)  # north_star: 1e-5 fp32
            assert np.array_equal(col[j].astype(np.float64), oo[b]['collisions']) and bool(dn[j]) == oo[b]['done'], (k, b)
    # distance-table reads counted by the kernel == the oracle's count for the sampled envs
    lk = _np(env.eng.t['lookups'][sample]).sum(axis=1)
    assert np.array_equal(lk, [oo[b]['lookups'] for b in ors])
    env.close()


def test_fullsize_determinism_and_env_independence():
    """65 536 envs, 4 steps: a second engine fed the same inputs in REVERSED env order
    returns the reversed outputs bit for bit (no cross-env coupling, no run-to-run noise)."""
    from red_gym_amd import workload
    B, A, T = 65536, 1, 4
    poses = torch.as_tensor(workload.spawn_poses(B, A))
    acts = torch.as_tensor(workload.action_pool(T, B, A))
    e1 = _mk(B, A, autoreset=True)
    e1.reset(poses.to(e1.device))
    for k in range(T):
        obs1, _, d1, i1 = e1.step(acts[k].to(e1.device))
    ref = {k: v.clone() for k, v in (('scans', obs1['scans']), ('state', e1.state), ('done', d1),
                                     ('col', obs1['collisions']), ('tog', i1['toggles']))}
    e1.close()
    e2 = _mk(B, A, autoreset=True)
    e2.reset(poses.flip(0).to(e2.device))
    for k in range(T):
        obs2, _, d2, i2 = e2.step(acts[k].flip(0).to(e2.device))
    assert torch.equal(obs2['scans'].flip(0), ref['scans'])
    assert torch.equal(e2.state.flip(0), ref['state'])
    assert torch.equal(d2.flip(0), ref['done']) and torch.equal(obs2['collisions'].flip(0), ref['col'])
    assert torch.equal(i2['toggles'].flip(0), ref['tog'])
    # sanity of the observation itself at this size
    s = ref['scans']
    assert bool(torch.isfinite(s).all()) and float(s.max()) <= 30.0 + 0.1 and float(s.min()) > -0.1
    e2.close()


def test_fullsize_autoreset_bookkeeping():
    """Random driving crashes most cars within a few hundred steps: every env that reported
    done is, one step later, back at its spawn pose with one zero-action step applied
    (time 0.01, noise row 1)."""
    from red_gym_amd import workload
    B, A = 65536, 1
    env = _mk(B, A, autoreset=True)
    poses = torch.as_tensor(workload.spawn_poses(B, A), device=env.device)
    acts = torch.as_tensor(workload.action_pool(8, B, A), device=env.device)
    env.reset(poses)
    total_done = 0
    for k in range(60):
        obs, _, done, info = env.step(acts[k % 8])
        was_done = done.clone()
        total_done += int(was_done.sum())
        if k % 10 == 9 and bool(was_done.any()):
            obs, _, done2, info = env.step(acts[(k + 1) % 8])
            idx = was_done.nonzero().flatten()
            ct = info['current_time'][idx]
            assert bool((ct == 0.01).all())
            assert bool((env.eng.t['noise_step'][idx, 0] == 1).all())
            assert bool((env.eng.t['toggles'][idx, 0] == 0).all())
            # x, y after reset + zero-action step == spawn x, y (the car does not move at v = 0)
            assert torch.equal(env.state[idx, 0, 0:2], poses[idx, 0, 0:2])
            total_done += int(done2.sum())
    assert total_done > B // 50
    env.close()


@pytest.mark.parametrize('B,A', [(16384, 1), (8192, 2)])
def test_scan_launch_order_changes_no_result(B, A):
    """f110_set_scan_order / Engine._reorder_scan: the wave that would march car i marches car order[i] -- a performance
    device (cars on the same noise row side by side).  Two engines, same inputs, autoreset on, 200 steps of random driving (the
    envs reset at different times, the sorted order is refreshed every 64 steps; the second engine additionally starts from a
    RANDOM permutation set through the ABI): every output tensor `==` the car-order run, bit for bit."""
    from red_gym_amd import workload
    poses = torch.as_tensor(workload.spawn_poses(B, A))
    acts = torch.as_tensor(workload.action_pool(8, B, A))
    e0 = _mk(B, A, autoreset=True, count_lookups=True)
    e0.eng.scan_reorder = False
    e1 = _mk(B, A, autoreset=True, count_lookups=True)
    e1.eng.REORDER_MIN_CARS = 1024        # (the engine sorts by itself from 65 536 cars on; here at every size)
    perm = torch.randperm(B * A, generator=torch.Generator().manual_seed(3)).to(dtype=torch.int32, device=e1.device)
    from red_gym_amd.engine import _lib, _ptr
    _lib.check(e1.eng.lib.f110_set_scan_order(e1.eng._h, _ptr(perm)))
    e0.reset(poses.to(e0.device)); e1.reset(poses.to(e1.device))
    for k in range(200):
        o0, _, d0, i0 = e0.step(acts[k % 8].to(e0.device))
        o1, _, d1, i1 = e1.step(acts[k % 8].to(e1.device))
        if k % 25 == 0 or k == 199:
            assert torch.equal(o0['scans'], o1['scans']) and torch.equal(e0.state, e1.state), k
            assert torch.equal(d0, d1) and torch.equal(o0['collisions'], o1['collisions']) and torch.equal(i0['toggles'], i1['toggles']), k
            assert torch.equal(e0.eng.t['lookups'], e1.eng.t['lookups']), k
    assert e1.eng._scan_order is not None and e1.eng._reorder_count > 128
    srt = torch.sort(e1.eng._scan_order.to(torch.int64)).values
    assert torch.equal(srt, torch.arange(B * A, device=e1.device))          # it is a permutation
    assert int(e1.eng.t['noise_step'].max()) > int(e1.eng.t['noise_step'].min()) + 50   # the envs really stand on different rows
    e0.close(); e1.close()
