"""The lidar noise on the GPU (csrc/f110_noise.h) and the per-env constructor arguments `seed` / `params`:
rows produced on the device against NumPy's own `default_rng(seed).normal` (reference call site laser_models.py:450-452,
seeding base_classes.py:117,202; golden g2 generated from the reference's rng use), K envs with K seeds and K vehicles
in ONE handle against K separate oracle envs, the ring (flat memory in a long run), re-production after resets, and the
device error word."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def _scanner(assets):
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    return sc


def _engine(assets, **kw):
    from red_gym_amd.engine import Engine
    e = Engine(**kw)
    e.set_map(os.path.join(assets, 'example_map.yaml'), '.png')
    return e


def test_device_rows_are_numpys_rows(assets, golden):
    """Rows 0..1299 of eight seeds `==` np.random.default_rng(seed).normal(0, 0.01, 1080) BIT FOR BIT, every draw: 11.2 M draws, of
    which ~2 900 come from the ziggurat's tail branch (|x| > 3.654: NumPy calls the C library's log1p there; the device restates
    glibc's log1p operation by operation, csrc/f110_noise.h log1p_glibc -- round 4 used the device math library and 4 of 2 156
    tail draws differed by an ulp).  The comparison is against the NumPy of the machine the test runs on (glibc 2.35 in this image).
    1 300 rows cross the first growth of the table (1 024 -> 2 048 rows per slot).  Seed 12345's first rows are also
    the reference-generated golden g2."""
    seeds = [12345, 0, 1, 7, 2 ** 31, 987654321987, 42, 2 ** 63 + 5]
    e = _engine(assets, num_envs=len(seeds), seed=seeds, noise_steps=1300)
    g2 = golden('g2_noise.npz')['seed12345']
    n_tail = n_draws = 0
    for k, sd in enumerate(seeds):
        got = e.noise_rows(k, 0, 1300)
        rng = np.random.default_rng(sd)
        ref = np.stack([rng.normal(0., 0.01, size=1080) for _ in range(1300)])
        tail = np.abs(ref) > 0.01 * 3.6541528853610088          # |x| > ziggurat_nor_r: produced by the log1p branch
        assert np.array_equal(got, ref), (sd, int((got != ref).sum()), int(tail.sum()))
        n_tail += int(tail.sum()); n_draws += ref.size
        if sd == 12345:
            assert np.array_equal(got[:g2.shape[0]], g2)
    assert n_draws >= 10 ** 7 and n_tail >= 2000, (n_draws, n_tail)   # 0.026 % of the draws
    print('draws: %d, tail draws: %d, all bit-identical to NumPy' % (n_draws, n_tail))
    assert e.device_errors() == 0
    e.close()


@pytest.mark.parametrize('source', ['device', 'numpy'])
def test_k_seeds_and_k_vehicles_are_k_reference_envs(assets, source):
    """One handle, 12 envs built with 4 seeds x 3 `params` dicts (f110_env.py:102-105,125-128), two agents, autoreset:
    every env `==` its own oracle Env constructed with that seed's noise rows and that params dict, 60 steps through
    wall hits and resets (state / scans 1e-9, collisions, toggles, done ==)."""
    import torch
    from red_gym_amd import F110VecEnv, workload
    from red_gym_amd.engine import DEFAULT_PARAMS
    B, A, T = 12, 2, 60
    seeds = [11, 22, 33, 44]
    pvar = [dict(DEFAULT_PARAMS), dict(DEFAULT_PARAMS, mu=0.7, C_Sf=3.9, m=4.5, a_max=7.0), dict(DEFAULT_PARAMS, mu=1.3, lf=0.17, lr=0.16, I=0.05, sv_max=2.5, width=0.29, length=0.55)]
    env_seed = [seeds[e % 4] for e in range(B)]
    env_par = [pvar[e % 3] for e in range(B)]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=A, seed=env_seed, params=env_par, autoreset=True,
                     keep_f64_scans=True, noise_source=source)
    sc = _scanner(assets)       # beam tables: the first env's params (class-level statics of the reference, base_classes.py:116-156)
    tables = {sd: oracle.noise_table(sd, T + 4) for sd in seeds}
    ors = [oracle.Env(sc, A, params=env_par[e], noise=tables[env_seed[e]]) for e in range(B)]
    poses = workload.spawn_poses(B, A)
    acts = workload.action_pool(8, B, A)
    acts[:, 3:7, :, 0] = 0.35
    acts[:, 3:7, :, 1] = 7.0     # some envs steer into the wall: iTTC hit, done, autoreset
    env.reset(poses)
    oo = [ors[e].reset(poses[e]) for e in range(B)]
    pend = np.zeros(B, dtype=bool)
    dones = 0
    for k in range(T):
        obs, _, done, info = env.step(torch.as_tensor(acts[k % 8], device='cuda'))
        for e in range(B):
            oo[e] = ors[e].reset(poses[e]) if pend[e] else ors[e].step(acts[k % 8][e])
        st, sc64 = _np(env.state), _np(obs['scans_f64'])
        for e in range(B):
            assert np.allclose(st[e], oo[e]['state'], rtol=0, atol=1e-9), (k, e)
            assert np.allclose(sc64[e], oo[e]['scans'], rtol=0, atol=1e-9), (k, e)
            assert np.array_equal(_np(obs['collisions'])[e].astype(np.float64), oo[e]['collisions']), (k, e)
            assert np.array_equal(_np(info['toggles'])[e].astype(np.float64), oo[e]['toggles']), (k, e)
            assert bool(_np(done)[e]) == oo[e]['done'], (k, e)
        pend = _np(done).astype(bool)
        dones += int(pend.sum())
    assert dones > 0
    # different seeds really are different streams, different vehicles different trajectories
    assert not np.array_equal(sc64[0], sc64[1]) and env.eng.device_errors() == 0
    env.close()


def test_every_env_its_own_seed_4096_envs_are_4096_reference_envs(assets):
    """`F110VecEnv(seed=list(...))` with MORE distinct seeds than the noise table has slots -- 4 096 envs,
    4 096 seeds (f110_env.py:102-105: every env is constructed with its own seed) -- runs with one generator per env
    (noise_source 'per_env', chosen by itself here), and every env `==` its own oracle Env fed NumPy's rows for that seed
    (default_rng(seed).normal, base_classes.py:117,202; laser_models.py:450-452), 24 steps through wall hits and autoresets
    (state / scans 1e-9; the noise itself bit for bit: a parked car's scan IS its row)."""
    import torch
    from red_gym_amd import F110VecEnv, workload
    B, A, T = 4096, 1, 24
    seeds = [1000 + 7 * e for e in range(B)]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=A, seed=seeds, autoreset=True, keep_f64_scans=True)
    assert env.eng._noise_per_env and env.eng.noise_info()[3] in (1, B)
    sc = _scanner(assets)
    ors = [oracle.Env(sc, A, noise=oracle.noise_table(sd, T + 4)) for sd in seeds]
    poses = workload.spawn_poses(B, A)
    poses[:8] = np.array([[-45.87478769831466, -16.282154624538293, 0.3]])      # inside a wall: the scan is the noise row
    acts = workload.action_pool(8, B, A)
    acts[:, 100:600, :, 0] = 0.35
    acts[:, 100:600, :, 1] = 7.0     # these steer into the wall: iTTC hit, done, autoreset -> their streams restart
    env.reset(poses)
    oo = [ors[e].reset(poses[e]) for e in range(B)]
    pend = np.zeros(B, dtype=bool)
    dones = 0
    for k in range(T):
        obs, _, done, info = env.step(torch.as_tensor(acts[k % 8], device='cuda'))
        st, sc64, dn = _np(env.state), _np(obs['scans_f64']), _np(done).astype(bool)
        for e in range(B):
            oo[e] = ors[e].reset(poses[e]) if pend[e] else ors[e].step(acts[k % 8][e])
        want_st = np.stack([o['state'] for o in oo]); want_sc = np.stack([o['scans'] for o in oo])
        assert np.allclose(st, want_st, rtol=0, atol=1e-9), k
        assert np.allclose(sc64, want_sc, rtol=0, atol=1e-9), k
        assert np.array_equal(sc64[:8], want_sc[:8]), k                 # noise rows: NumPy's bits
        assert np.array_equal(dn, np.array([o['done'] for o in oo])), k
        pend = dn
        dones += int(pend.sum())
    assert dones > 100 and env.eng.device_errors() == 0
    assert not np.array_equal(sc64[0], sc64[1])
    env.close()


def test_per_env_noise_follows_a_loaded_checkpoint_and_a_masked_reset(assets):
    """Per-env generators keep no table: a row counter that does not continue the generator's (state restored from a
    checkpoint) makes the generator run forward from its seed; a masked reset restarts only the masked envs' streams."""
    import torch
    from red_gym_amd import F110VecEnv
    B = 6
    seeds = [3, 4, 5, 3, 4, 5]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, seed=seeds, autoreset=False, keep_f64_scans=True,
                     noise_source='per_env')
    poses = np.tile(np.array([[[-45.87478769831466, -16.282154624538293, 0.3]]]), (B, 1, 1))   # inside a wall
    z = torch.zeros((B, 1, 2), dtype=torch.float64, device='cuda')
    env.reset(poses)
    for k in range(40):
        env.step(z)
    sd = env.state_dict()
    for k in range(5):
        env.step(z)
    s45 = _np(env.eng.t['scans_f64']).copy()
    env.load_state_dict(sd)                      # back to row 41: the generators stand at 46
    for k in range(5):
        env.step(z)
    assert np.array_equal(_np(env.eng.t['scans_f64']), s45)
    rows = {s_: np.random.default_rng(s_).normal(0., 0.01, size=(60, 1080)) for s_ in (3, 4, 5)}
    assert np.array_equal(s45[:, 0], np.stack([rows[s_][45] for s_ in seeds]))
    m = torch.zeros(B, dtype=torch.uint8, device='cuda'); m[1] = 1; m[5] = 1
    env.reset(poses, m)
    got = _np(env.eng.t['scans_f64'])[:, 0]
    assert np.array_equal(got[1], rows[4][0]) and np.array_equal(got[5], rows[5][0])
    env.step(z)
    got = _np(env.eng.t['scans_f64'])[:, 0]
    want = np.stack([rows[seeds[e]][1 if e in (1, 5) else 46] for e in range(B)])
    assert np.array_equal(got, want) and env.eng.device_errors() == 0
    env.close()


def test_update_params_reaches_every_slot_and_one_slot_only(assets):
    """f110_update_params (Simulator.update_params, base_classes.py:507-527) applies to the agent in EVERY env;
    f110_set_params_slot(slot, agent) to that env's agent only; IndexError beyond the agent list."""
    import torch
    from red_gym_amd import F110VecEnv, _lib, workload
    from red_gym_amd.engine import DEFAULT_PARAMS, params_vec
    B, A = 4, 2
    pa, pb, pc = dict(DEFAULT_PARAMS), dict(DEFAULT_PARAMS, mu=0.6), dict(DEFAULT_PARAMS, mu=1.4, a_max=5.0)
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=A, params=[pa, pb, pa, pb], autoreset=False,
                     keep_f64_scans=True, noise_std=0)
    env.update_params(pc, index=1)                                     # agent 1 of every env
    v = params_vec(pb); v[0] = 0.9                                     # agent 0 of the envs on slot 1 only
    _lib.check(env.eng.lib.f110_set_params_slot(env.eng._h, 1, v.ctypes.data_as(C.c_void_p), 0))
    with pytest.raises(IndexError):
        env.update_params(pc, index=2)
    sc = _scanner(assets)
    poses = workload.spawn_poses(B, A)
    acts = workload.action_pool(8, B, A)
    env.reset(poses)
    # the oracle's Env holds ONE params vector for all its agents: drive each agent in an env of its own kind
    p_agent = lambda e, a: pc if a == 1 else (dict(pb, mu=0.9) if e % 2 == 1 else pa)  # noqa: E731
    solo = [[oracle.Env(sc, 1, params=p_agent(e, a)) for a in range(A)] for e in range(B)]
    for e in range(B):
        for a in range(A):
            solo[e][a].reset(poses[e, a:a + 1])
    for k in range(25):
        env.step(torch.as_tensor(acts[k % 8], device='cuda'))
        st = _np(env.state)
        for e in range(B):
            for a in range(A):
                assert np.allclose(st[e, a], solo[e][a].step(acts[k % 8][e, a:a + 1])['state'][0], rtol=0, atol=1e-9), (k, e, a)
    env.close()


def test_long_run_is_a_ring_of_constant_size(assets):
    """20 000 steps, autoreset off, replayed from ONE captured hipGraph: the noise table stays at its first size (a ring
    following the cars), the launch epoch never moves (no re-capture), no device error, and the scans are NumPy's rows.
    The cars sit inside a wall (the first table read is 0: every beam returns 0 + noise), so a scan IS its noise row."""
    import torch
    from red_gym_amd import F110VecEnv
    B, T = 4, 20000
    seeds = [5, 6, 5, 6]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, seed=seeds, autoreset=False, keep_f64_scans=True)
    poses = np.tile(np.array([[[-45.87478769831466, -16.282154624538293, 0.3]]]), (B, 1, 1))   # the centre of a wall cell (map row 449, col 517)
    env.reset(poses)
    buf = env.capture_step()
    buf.zero_()
    ep0, bytes0 = env.eng.launch_epoch(), env.eng.noise_info()[4]
    rows = {sd: np.random.default_rng(sd) for sd in (5, 6)}
    ref = {sd: rows[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}  # row 0 went to the reset's scan
    check_at = set(range(0, T, 997)) | {1023, 1024, 1025, 2047, 2048, T - 1}
    for k in range(T):
        env.step_graph()
        ref = {sd: rows[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}
        if k in check_at:
            s = _np(env.eng.t['scans_f64'])
            for e in range(B):
                r = ref[seeds[e]]
                tail = np.abs(r) > 0.01 * 3.6541528853610088
                assert np.array_equal(s[e, 0][~tail], r[~tail]), (k, e)
                assert np.allclose(s[e, 0], r, rtol=0, atol=2e-17), (k, e)
    lo, hi, cap, slots, nbytes = env.eng.noise_info()
    assert env.eng.launch_epoch() == ep0 and nbytes == bytes0 and cap == 1024 and slots == 2
    assert lo > T - 1024 - 300 and hi > T and env.eng.device_errors() == 0
    assert int(env.eng.t['noise_step'].min()) == T + 1
    # a masked reset sends two cars back to row 0 while the others run on: the dropped rows are produced again
    m = torch.zeros(B, dtype=torch.uint8, device='cuda'); m[:2] = 1
    env.reset(poses, m)
    rr = {sd: np.random.default_rng(sd) for sd in (5, 6)}
    first = {sd: rr[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}
    s = _np(env.eng.t['scans_f64'])
    for e in (0, 1):
        assert np.allclose(s[e, 0], first[seeds[e]], rtol=0, atol=2e-17), e
    env.step(torch.zeros((B, 1, 2), dtype=torch.float64, device='cuda'))
    s = _np(env.eng.t['scans_f64'])
    nxt = {sd: rows[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}
    sec = {sd: rr[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}
    for e in range(B):
        assert np.allclose(s[e, 0], (sec if e < 2 else nxt)[seeds[e]], rtol=0, atol=2e-17), e
    assert env.eng.device_errors() == 0 and env.eng.noise_info()[2] >= 32768   # the ring now spans rows 0 .. 20 002
    env.close()


def test_prefetch_waits_for_the_steps_enqueued_before_the_floor_moved(assets):
    """the floor of the noise ring is raised from the HOST's step count, and the next prefetch (the library's
    side stream) recycles the ring places of the rows below it -- while steps that were enqueued before may not have run yet
    (a 65 536-env shard runs hundreds of steps behind its host).  The trace of Engine._ensure_noise, compressed: the stream is
    stalled by a sleep kernel, 300 steps are enqueued behind it (rows 1 .. 300), the floor goes to 301 and a prefetch of rows
    1 024 .. 1 279 is started -- ring places 0 .. 255.  Every one of the 300 scans must still be its own NumPy row (the cars sit
    inside a wall: a scan IS its noise row).  Fails on the round-4 library (the side stream did not wait)."""
    import ctypes as C
    import torch
    from red_gym_amd import F110VecEnv
    B, T = 4, 300
    seeds = [5, 6, 5, 6]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, seed=seeds, autoreset=False, keep_f64_scans=True)
    poses = np.tile(np.array([[[-45.87478769831466, -16.282154624538293, 0.3]]]), (B, 1, 1))
    env.reset(poses)
    eng = env.eng
    eng._noise_to(1024)
    torch.cuda.synchronize()
    lo, hi, cap = eng.noise_info()[:3]
    assert (lo, hi, cap) == (0, 1024, 1024)
    z = torch.zeros((B, 1, 2), dtype=torch.float64, device='cuda')
    keep = torch.empty((T, B, 1080), dtype=torch.float64, device='cuda')
    torch.cuda._sleep(int(1.2e9))            # ~0.5 s: everything below is enqueued behind it
    for k in range(T):
        env.step(z)
        keep[k].copy_(eng.t['scans_f64'][:, 0])
    lib = eng.lib
    assert lib.f110_noise_set_floor(eng._h, C.c_int64(T + 1), eng._stream()) == 0
    assert lib.f110_noise_prefetch(eng._h, C.c_int64(1024 + 256)) == 0
    torch.cuda.synchronize()
    got = keep.cpu().numpy()
    rows = {sd: np.random.default_rng(sd) for sd in (5, 6)}
    for sd in (5, 6):
        rows[sd].normal(0., 0.01, size=1080)  # row 0 went to the reset's scan
    for k in range(T):
        ref = {sd: rows[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}
        for e in range(B):
            assert np.allclose(got[k, e], ref[seeds[e]], rtol=0, atol=2e-17), (k, e)
    assert eng.device_errors() == 0
    env.close()


def test_masked_reset_after_a_long_run_reproduces_dropped_rows_from_the_marks(assets):
    """a masked reset after the floor has moved used to rewind every generator to its seed and re-run the
    whole stream (one wavefront, ~15 us per row: 0.3 s at 20 000 steps).  The dropped rows are now produced again from the
    marks the generators leave every 64 rows, one wavefront per 64 rows: the generators' states do not move, the rows are
    NumPy's, and the reset takes milliseconds."""
    import time
    import torch
    from red_gym_amd import F110VecEnv
    B, T = 4, 6000
    seeds = [5, 6, 5, 6]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, seed=seeds, autoreset=False, keep_f64_scans=True)
    poses = np.tile(np.array([[[-45.87478769831466, -16.282154624538293, 0.3]]]), (B, 1, 1))
    env.reset(poses)
    z = torch.zeros((B, 1, 2), dtype=torch.float64, device='cuda')
    for k in range(T):
        env.step(z)
    torch.cuda.synchronize()
    lo0, hi0 = env.eng.noise_info()[:2]
    assert lo0 > T - 1024 - 300
    m = torch.zeros(B, dtype=torch.uint8, device='cuda'); m[:2] = 1
    t0 = time.perf_counter()
    env.reset(poses, m)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lo1, hi1 = env.eng.noise_info()[:2]
    assert lo1 == 0 and hi1 >= hi0           # nothing that had been produced was thrown away
    # every row of the table is NumPy's (rows 0 .. lo0-1 come from the marks, the rest from the first pass)
    for sl, sd in ((0, 5), (1, 6)):
        rng = np.random.default_rng(sd)
        want = rng.normal(0., 0.01, size=(hi0, 1080))
        got = env.eng.noise_rows(sl, 0, hi0)
        assert np.allclose(got, want, rtol=0, atol=2e-17), sd
        tail = np.abs(want) > 0.01 * 3.6541528853610088
        assert np.array_equal(got[~tail], want[~tail]), sd
    # and the run goes on for both cohorts
    rr = {sd: np.random.default_rng(sd) for sd in (5, 6)}
    for sd in (5, 6):
        rr[sd].normal(0., 0.01, size=1080)
    for k in range(3):
        env.step(z)
        s = _np(env.eng.t['scans_f64'])
        sec = {sd: rr[sd].normal(0., 0.01, size=1080) for sd in (5, 6)}
        for e in (0, 1):
            assert np.allclose(s[e, 0], sec[seeds[e]], rtol=0, atol=2e-17), (k, e)
    assert env.eng.device_errors() == 0
    assert dt < 0.08, 'masked reset after %d steps took %.3f s' % (T, dt)   # (~90 ms to re-run 6 000 rows serially; a few ms now)
    env.close()


def test_autoreset_keeps_every_row_and_grows_quietly(assets):
    """autoreset on: a car may be sent back to row 0 at any step, so rows are never dropped; a car that outlives the table
    makes it double (1 024 -> 2 048 rows).  The re-allocation is the one event that moves the launch epoch (the scan takes
    the table's base by value): a captured graph is re-captured by step_graph and keeps giving the eager results."""
    import torch
    from red_gym_amd import F110VecEnv
    B = 2
    e1 = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, autoreset=True, keep_f64_scans=True)
    e2 = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=1, autoreset=True, keep_f64_scans=True, noise_source='numpy')
    poses = np.array([[[0.7, 0.0, 1.37079632679]], [[0.0, 20.0, 0.3]]])
    e1.reset(poses); e2.reset(poses)
    buf = e1.capture_step()
    buf.zero_()
    ep0 = e1.eng.launch_epoch()
    z = torch.zeros((B, 1, 2), dtype=torch.float64, device='cuda')
    for k in range(1100):
        e1.step_graph(); e2.step(z)
        if k % 100 == 99 or k > 1015:
            assert torch.allclose(e1.eng.t['scans_f64'], e2.eng.t['scans_f64'], rtol=0, atol=2e-17), k
    assert e1.eng.launch_epoch() == ep0 + 1 and e1.eng.noise_info()[0] == 0 and e1.eng.noise_info()[2] == 2048
    assert e1.eng.device_errors() == 0 and e2.eng.device_errors() == 0
    e1.close(); e2.close()


def test_a_row_outside_the_table_is_reported(assets):
    """The raw ABI stepped past the rows it was told to produce: F110_DEVERR_NOISE_WINDOW in the device error word."""
    import torch
    e = _engine(assets, num_envs=2, noise_steps=64)
    poses = torch.as_tensor(np.array([[[0.7, 0.0, 1.37]], [[0.7, 0.0, 1.37]]]), device='cuda')
    e.reset(poses)
    assert e.device_errors() == 0
    e.t['noise_step'].fill_(5000)           # as if the cars had made 5 000 scans nobody announced
    a = torch.zeros((2, 1, 2), dtype=torch.float64, device='cuda')
    with torch.cuda.device(e.device):
        assert e.lib.f110_step(e._h, C.c_void_p(a.data_ptr()), None) == 0
    assert e.device_errors() == 1 and e.device_errors() == 0      # reported once, then cleared
    e.close()


def test_every_env_its_own_vehicle_at_batch_scale(assets):
    """4 096 envs, each constructed with its own params dict (4 096 params slots, identity assignment) and one of 64 seeds:
    a sample of envs spread over the batch `==` their own oracle Envs over 12 steps (what BASELINE's domain-randomised use
    looks like: SURVEY 8 f3)."""
    import torch
    from red_gym_amd import F110VecEnv, workload
    from red_gym_amd.engine import DEFAULT_PARAMS
    B, A, T = 4096, 1, 12
    rng = np.random.default_rng(99)
    pars = []
    for e in range(B):
        p = dict(DEFAULT_PARAMS)
        p.update(mu=float(rng.uniform(0.6, 1.3)), C_Sf=float(rng.uniform(3.5, 5.5)), C_Sr=float(rng.uniform(4.0, 6.0)),
                 m=float(rng.uniform(3.0, 4.5)), I=float(rng.uniform(0.035, 0.06)), a_max=float(rng.uniform(6.0, 10.0)))
        pars.append(p)
    seeds = [1000 + (e * 7) % 64 for e in range(B)]
    env = F110VecEnv(B, map=os.path.join(assets, 'example_map'), num_agents=A, seed=seeds, params=pars, autoreset=False, keep_f64_scans=True)
    poses = workload.spawn_poses(B, A)
    acts = workload.action_pool(4, B, A)
    env.reset(poses)
    sample = [0, 1, 63, 64, 65, 1000, 2047, 2048, 3000, 4095]
    sc = _scanner(assets)
    ors = {e: oracle.Env(sc, A, params=pars[e], noise=oracle.noise_table(seeds[e], T + 3)) for e in sample}
    for e in sample:
        ors[e].reset(poses[e])
    for k in range(T):
        obs = env.step(torch.as_tensor(acts[k % 4], device='cuda'))[0]
        st, s64 = _np(env.state), _np(obs['scans_f64'])
        for e in sample:
            o = ors[e].step(acts[k % 4][e])
            assert np.allclose(st[e], o['state'], rtol=0, atol=1e-9), (k, e)
            assert np.allclose(s64[e], o['scans'], rtol=0, atol=1e-9), (k, e)
    assert env.eng.device_errors() == 0
    env.close()


def test_slot_api_refuses_what_it_cannot_do(assets):
    """Error behaviour of the per-env slot calls: index errors for slots that do not exist, value errors for tables that
    cannot serve -- never a silent fallback."""
    import torch
    from red_gym_amd import F110VecEnv, _lib
    from red_gym_amd.engine import DEFAULT_PARAMS, Engine, params_vec
    with pytest.raises(ValueError):
        Engine(num_envs=100, seed=list(range(100)), noise_source='numpy')  # 100 distinct seeds > 64 slots of host rows
    e100 = Engine(num_envs=100, seed=list(range(100)))                    # ... the device switches to one generator per env
    assert e100._noise_per_env
    e100.close()
    with pytest.raises(ValueError):
        Engine(num_envs=4, seed=[1, 2, 3])                                # one seed per env
    with pytest.raises(ValueError):
        Engine(num_envs=4, params=[dict(DEFAULT_PARAMS)] * 3)             # one params dict per env
    e = _engine(assets, num_envs=4, seed=[7, 8, 7, 8])
    lib, h = e.lib, e._h
    v = params_vec(DEFAULT_PARAMS)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert lib.f110_set_params_slot(h, 4, P(v), -1) == _lib.E_INDEX       # slots 0 .. num_envs-1
    assert lib.f110_set_params_slot(h, 1, P(v), 1) == _lib.E_INDEX        # agent 1 of a one-agent handle
    bad = v.copy(); bad[3] = np.nan
    assert lib.f110_set_params_slot(h, 1, P(bad), -1) == _lib.E_INVALID
    assert lib.f110_set_params_slots(h, P(np.tile(v, 5)), 5) == _lib.E_INDEX
    assert lib.f110_set_params_slot(h, 2, P(v), -1) == 0                  # slots 1, 2 now exist (copies of slot 0, then set)
    assign = np.array([0, 1, 2, 3], dtype=np.int32)
    assert lib.f110_assign_params(h, P(assign)) == _lib.E_INDEX           # slot 3 does not
    assign[3] = 2
    assert lib.f110_assign_params(h, P(assign)) == 0
    na = np.array([0, 1, 2, 0], dtype=np.int32)
    assert lib.f110_assign_noise(h, P(na)) == _lib.E_INDEX                # two noise slots only
    assert lib.f110_noise_set_floor(h, 10 ** 6, None) == _lib.E_INVALID   # above the rows produced
    assert lib.f110_pack_env(h, 4, C.c_void_p(e.t['state'].data_ptr()), None) in (_lib.E_INDEX, _lib.E_UNBOUND)
    out = np.empty((2, 1080))
    assert lib.f110_noise_read(h, 0, 10 ** 6, 2, P(out)) == _lib.E_INDEX  # rows that are not in the table
    assert b'noise_read' in lib.f110_last_error()
    e.close()
    # a host table that is too short is an error at f110_noise_ensure, not a wrap-around
    e = _engine(assets, num_envs=1, noise_source='numpy')
    with torch.cuda.device(e.device):
        assert e.lib.f110_noise_ensure(e._h, 10 ** 6, None) == _lib.E_INVALID
        assert e.lib.f110_noise_set_floor(e._h, 1, None) == _lib.E_INVALID    # rows are only dropped from generated noise
    e.close()
