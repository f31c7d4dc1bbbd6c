#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched F110Env.step path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs B] [--agents A]

One "step" = one f110_step over all B envs of this rank (every car: single-track
RK4, 1080-beam ray march, noise, iTTC, (GJK + opponent ray cast for A>1), lap
logic, autoreset).  Default workload: BASELINE.json configs[2] = 65536 envs x 1
agent per GPU on example_map (the config the metric's target is quoted on; N>1 is
configs[4], independent shards, no collective on the step path).  Actions and all
state are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def usable_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup
    CPU quota (a GPU box hands a 1-GPU job a slice of a many-core host)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, 64)


def cpu_baseline(num_agents, budget_s=12.0):
    """The CPU oracle (oracle/f110_oracle.c, a scalar fp64 port of the reference's
    Numba path) timed on this box's host cores on a bounded sample of the same
    workload: same map, spawn distribution, action distribution, noise, autoreset."""
    import oracle
    from red_gym_amd import workload
    cores = usable_cores()
    sc = oracle.Scanner(1080, 2 * np.pi)
    sc.set_map(workload.EXAMPLE_MAP + '.yaml', '.png')
    B = 1024   # the same sample whatever the box (round 4 scaled it with the core count)
    noise = oracle.noise_table(12345, 4096)
    batch = oracle.Batch(sc, B, num_agents, workload.spawn_poses(B, num_agents), noise=noise)
    acts = workload.action_pool(8, B, num_agents)
    batch.step(acts[0], threads=cores)  # reset step (untimed)
    t0 = time.perf_counter()
    steps = 0
    while True:
        batch.step(acts[steps % 8], threads=cores)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 400:
            break
    dt = time.perf_counter() - t0
    # single-thread figure (the reference itself is single-threaded) on a smaller sample
    b1 = oracle.Batch(sc, 64, num_agents, workload.spawn_poses(64, num_agents), noise=noise)
    a1 = workload.action_pool(4, 64, num_agents)
    b1.step(a1[0], threads=1)
    t1 = time.perf_counter()
    for k in range(6):
        b1.step(a1[k % 4], threads=1)
    dt1 = time.perf_counter() - t1
    return {'value': B * steps / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d envs x %d agent(s) x %d steps (OpenMP over envs, %d threads), example_map, '
                      'same spawn/action/noise distribution' % (B, num_agents, steps, cores),
            'single_thread_value': 64 * 6 / dt1}


class _stdout_to_stderr(object):
    """Gloo announces its connections on the C-level stdout; rank 0's stdout carries the ONE JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)


class Ranks(object):
    """One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run).
    The step path has no collective: ranks only meet at the barriers around the timed
    region, to take the MAX of the elapsed time and to gather every rank's own time and
    device name for the JSON line.  backend 'nccl' (= RCCL) on GPUs, 'gloo' in the CPU tests.
    force_group: build the process group even for ONE rank (tests/test_gpu_rccl.py: RCCL init,
    barrier(device_ids), all_reduce(MAX), all_gather and destroy then run on real hardware on a
    one-GPU box -- the same calls the 8-GPU run makes)."""

    def __init__(self, backend='nccl', device=None, force_group=False):
        import torch.distributed as dist
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.backend, self.device, self.dist = backend, device, dist
        self.grouped = self.world > 1 or bool(force_group)
        self.last_per_rank = None
        if self.grouped:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:  # (only the forced one-rank group comes without a launcher)
                import socket
                with socket.socket() as s:
                    s.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(s.getsockname()[1])
            self._init_group(backend, device)

    GROUP_TIMEOUT_S = 180   # rendezvous, every collective of the bench and the backend vote: a dead rank ends the run

    def _init_group(self, backend, device):
        """One TCPStore (rank 0 hosts it at MASTER_ADDR:MASTER_PORT) carries everything: the process group's rendezvous,
        the ranks' VOTE on the backend, and -- if RCCL cannot start on ANY rank -- the gloo group that replaces it.  Every
        rank tries the asked-for backend (a probe all_reduce makes RCCL create its communicators here, where a failure can
        still be handled), publishes ok / failed under its own key and reads everybody's: the ranks fall back TOGETHER or
        not at all (round 4 let each rank decide for itself: a rank whose probe had succeeded stayed in the RCCL group and
        the others waited for it in a gloo barrier for ever).  The ranks only meet for barriers and two tiny gathers
        around the timed region -- no collective on the step path -- so the measurement is the same over gloo; the JSON
        line says which backend synchronised the ranks.  Every wait has a timeout."""
        import datetime
        dist = self.dist
        tmo = datetime.timedelta(seconds=self.GROUP_TIMEOUT_S)
        # Under torch.distributed.run the launcher's agent already HOSTS the store at MASTER_ADDR:MASTER_PORT
        # (TORCHELASTIC_USE_AGENT_STORE=True, what torch's own env:// rendezvous looks at): every rank is then a client of it;
        # self-launched ranks (python bench.py --gpus N) have no agent, and rank 0 hosts the store.
        agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
        store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), self.world,
                              is_master=(self.rank == 0 and not agent_store), timeout=tmo, wait_for_workers=False)
        if agent_store:   # (keys of an earlier attempt of the same launcher must not be seen)
            store = dist.PrefixStore('f110_bench/attempt_%s' % os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), store)
        self._store = store
        fail_ranks = [int(r) for r in os.environ.get('F110_BENCH_TEST_PRIMARY_FAILS', '').split(',') if r.strip()]  # tests only
        err = None
        try:
            with _stdout_to_stderr():
                kw = {'device_id': device} if backend == 'nccl' else {}
                dist.init_process_group(backend, store=dist.PrefixStore('primary', store), rank=self.rank, world_size=self.world,
                                        timeout=tmo, **kw)
            if self.rank in fail_ranks:
                raise RuntimeError('injected failure of the primary backend (F110_BENCH_TEST_PRIMARY_FAILS)')
            if backend == 'nccl':
                import torch
                probe = torch.zeros(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize(device)
        except Exception as e:  # noqa: BLE001 -- whatever the backend raises on this node
            err = e
            sys.stderr.write('bench.py: rank %d: backend %s failed to start (%s: %s)\n' % (self.rank, backend, type(e).__name__, e))
        store.set('vote/%d' % self.rank, b'0' if err else b'1')
        store.wait(['vote/%d' % r for r in range(self.world)], tmo)
        all_ok = all(store.get('vote/%d' % r) == b'1' for r in range(self.world))
        if all_ok:
            with _stdout_to_stderr():
                dist.barrier(device_ids=[device.index]) if backend == 'nccl' else dist.barrier()
            return
        if backend == 'gloo' and not fail_ranks:
            raise RuntimeError('bench.py: the gloo process group could not be built: %r' % (err,))
        if self.rank == 0:
            sys.stderr.write('bench.py: backend %s did not start on every rank; all %d ranks synchronise over gloo\n' % (backend, self.world))
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        self.backend = 'gloo'
        with _stdout_to_stderr():
            dist.init_process_group('gloo', store=dist.PrefixStore('fallback', store), rank=self.rank, world_size=self.world, timeout=tmo)
            dist.barrier()

    def _comm_device(self):
        return self.device if self.backend == 'nccl' else 'cpu'

    def sync_device(self):
        if self.device is not None:
            import torch
            torch.cuda.synchronize(self.device)

    def barrier(self):
        self.sync_device()
        if self.grouped:
            if self.backend == 'nccl':
                self.dist.barrier(device_ids=[self.device.index])
            else:
                self.dist.barrier()
        self.sync_device()

    def max_over_ranks(self, seconds, own=None):
        """MAX over ranks of `seconds` (the time between the two barriers).  `own` = this rank's time until ITS OWN work
        was done, taken before the closing barrier: gathered from every rank into last_per_rank, so that a straggler
        shows (after the barrier all ranks read the same clock)."""
        own = seconds if own is None else own
        if not self.grouped:
            self.last_per_rank = [own]
            return seconds
        import torch
        t = torch.tensor([own], dtype=torch.float64, device=self._comm_device())
        each = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(each, t)
        self.last_per_rank = [float(x.item()) for x in each]
        t = torch.tensor([seconds], dtype=torch.float64, device=self._comm_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_names(self, name):
        """Every rank's device name (fixed 64-byte tensors: no pickling over the collective backend)."""
        if not self.grouped:
            return [name]
        import torch
        raw = name.encode('utf-8', 'replace')[:64].ljust(64, b'\0')
        t = torch.tensor(list(raw), dtype=torch.uint8, device=self._comm_device())
        each = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(each, t)
        return [bytes(x.cpu().tolist()).rstrip(b'\0').decode('utf-8', 'replace') for x in each]

    def close(self):
        if self.grouped:
            self.barrier()
            self.dist.destroy_process_group()


def timed_steps(ranks, step_fn, K):
    """EXACTLY K steps bracketed by barrier + device sync on both sides; returns the MAX
    over ranks of the elapsed seconds."""
    ranks.barrier()
    t0 = time.perf_counter()
    for k in range(K):
        step_fn(k)
    ranks.sync_device()
    own = time.perf_counter() - t0      # this rank's own K steps (per_rank_ms); not what `value` is computed from
    ranks.barrier()
    return ranks.max_over_ranks(time.perf_counter() - t0, own)


def rank_workload(rank, B, A, pool=16):
    """Each rank owns an independent shard of envs: its own spawn jitter (seed 2025+rank)
    and action stream (seed 777+rank); SURVEY 8(d)."""
    from red_gym_amd import workload
    return workload.spawn_poses(B, A, rank), workload.action_pool(pool, B, A, rank)


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N FRESH rank processes of this file (one per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment -- what torch.distributed.run
    would set), wait for them and hand back the first non-zero exit code.  The parent never imports torch
    or touches the GPU and nothing is re-exec'd; rank 0's child prints the one JSON line on the inherited
    stdout.  If a rank dies the others (which would wait at a barrier for ever) are terminated."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write('bench.py: rank %d exited with code %d; stopping the other ranks\n' % (r, code))
                    for q in live:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        # the launcher itself is being stopped (timeout, Ctrl-C): do not leave ranks behind
        for r in live:
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except Exception:
                procs[r].kill()
    return rc


def stub_bench(args):
    """Rehearsal of the rank plumbing WITHOUT a GPU (tests/test_bench_dist_cpu.py): gloo ranks, the real
    barriers / MAX-over-ranks / JSON assembly, and a sleep of F110_BENCH_STUB_STEP_MS (x (1 + rank), so that
    the slowest rank sets the time) standing in for f110_step.  The line says "data": "stub": it is not a
    measurement and the driver's runs never set the variable."""
    ranks = Ranks('gloo', None)
    ms = float(os.environ['F110_BENCH_STUB_STEP_MS'])
    poses, acts = rank_workload(ranks.rank, args.envs, args.agents, 2)
    n = []

    def step_fn(k):
        n.append(k)
        time.sleep(ms * 1e-3 * (1 + ranks.rank))
    for k in range(args.warmup):
        step_fn(k)
    del n[:]
    elapsed = timed_steps(ranks, step_fn, args.steps)
    assert len(n) == args.steps
    per_rank = list(ranks.last_per_rank)
    names = ranks.gather_names('stub-cpu-%d' % ranks.rank)
    world = ranks.dist.get_world_size() if ranks.grouped else 1
    out = {'metric': 'env steps/sec (all envs), 1080-beam lidar', 'value': world * args.envs * args.steps / elapsed,
           'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'stub',
           'per_rank_ms': [t / args.steps * 1e3 for t in per_rank], 'devices': names,
           'config': {'workload': 'STUB step (sleep), rank plumbing only', 'envs_per_gpu': args.envs,
                      'agents': args.agents, 'shard_checksum': float(poses.sum() + acts.sum())},
           'roofline': None}
    rank = ranks.rank
    ranks.close()
    if rank == 0:
        print(json.dumps(out), flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--agents', type=int, default=1)
    ap.add_argument('--policy', choices=['random', 'pure_pursuit'], default='random',
                    help='random actions (the headline workload, SURVEY 8d) or closed-loop pure pursuit on the GPU')
    ap.add_argument('--bitmap', choices=['none', 'FILL', 'POLYGON', 'RAYS'], default='none',
                    help="secondary mode: also draw every ego scan to a 256x256 bird's-eye bitmap after each step "
                         '(lidar_to_bitmap, what the RL consumers do with the scans)')
    ap.add_argument('--spinup', type=int, default=0,
                    help='EXTRA untimed steps before reset + warmup (an idle MI355X needs ~25 ms of load to reach its '
                         'clocks: tools/step_ramp.py).  0 (default): the protocol is exactly W warm-up + K timed steps; '
                         'the effect of the clock ramp is reported separately as `sustained`')
    ap.add_argument('--sustained', type=int, default=200,
                    help='after the timed region, time this many further steps (clocks up, cars scattered) and report '
                         'them as `sustained` beside `value` (0 = skip); never part of `value`')
    ap.add_argument('--steady-state', type=int, default=200,
                    help='after `sustained`, stagger the envs over 1 024 noise rows (the regime of a long-running batch) and time '
                         'this many steps; reported as `steady_state` beside `value` (0 = skip); never part of `value`')
    ap.add_argument('--repeats', type=int, default=1,
                    help='R > 1: after the protocol region (which alone gives `value`), R - 1 further timed regions of K steps '
                         'each; the median of all R is reported beside it as `median_of_repeats` (SURVEY 8d: median of 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-scan-events', action='store_true',
                    help='do not bracket the scan kernel with hipEvents (roofline becomes null)')
    ap.add_argument('--scan-events-every', type=int, default=4,
                    help='hipEvent pairs ride on the scan dispatch of every Nth TIMED step (a dispatch with events costs '
                         'the stream ~10 us of idle time around it); 1 = every step')
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    # N > 1 without a launcher (the driver's `python3 bench.py --gpus N` form): start the ranks ourselves, BEFORE
    # torch is imported or any HIP call is made in this process
    if args.gpus > 1 and 'RANK' not in os.environ:
        return self_launch(args.gpus, argv)
    if os.environ.get('F110_BENCH_STUB_STEP_MS'):
        return stub_bench(args)

    import torch
    from red_gym_amd import F110VecEnv, workload

    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Rehearsal hooks for a one-GPU box (the driver's runs use neither): F110_BENCH_ONE_DEVICE=1 puts every
    # rank on device 0 and F110_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device.
    if os.environ.get('F110_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    if local_rank >= torch.cuda.device_count():  # counting devices does not initialise the GPU
        sys.exit('bench.py: rank with LOCAL_RANK=%d but only %d GPU(s) are visible' % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # F110_BENCH_FORCE_GROUP=1 (tests/test_gpu_rccl.py): a ONE-rank RCCL process group, so that the collective calls of the
    # multi-GPU run execute on a one-GPU box
    ranks = Ranks(os.environ.get('F110_BENCH_BACKEND', 'nccl'), dev, force_group=os.environ.get('F110_BENCH_FORCE_GROUP') == '1')
    rank = ranks.rank
    world = ranks.dist.get_world_size() if ranks.grouped else 1  # as the process group reports it
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d\n'
                         % (args.gpus, world, world))

    B, A, K, W = args.envs, args.agents, args.steps, args.warmup
    POOL = 16
    # host-side preparation first (~0.1 s of NumPy), so that everything the GPU does before the timed region -- map
    # pipeline, uploads, reset, warm-up -- follows back to back instead of being separated from it by an idle stretch
    poses_np, acts_np = rank_workload(rank, B, A, POOL)
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, map_ext='.png', num_agents=A, timestep=0.01, seed=12345,
                     device=local_rank, autoreset=True, count_lookups=True)
    poses = torch.as_tensor(poses_np, device=dev)
    acts = torch.as_tensor(acts_np, device=dev)  # resident in HBM before the timed region
    # The first ~30 steps after an idle period run up to 15 % slower (0.80 -> 0.69 ms) whatever the env state
    # (profiles/r02_step_ramp.txt: the GPU leaving its idle power state), so a short run (--steps 20 --warmup 5)
    # measures mostly that ramp.  `value` is nevertheless the protocol as asked -- reset, W warm-up steps, K timed
    # steps, nothing else -- and the figure with the clocks up is reported beside it (`sustained`, below).
    # --spinup N > 0 (not the default) runs N throw-away steps first, for experiments.
    if args.spinup > 0:
        env.reset(poses)
        for k in range(args.spinup):
            env.step(acts[k % POOL])
    env.reset(poses)
    for k in range(W):
        env.step(acts[k % POOL])
    lookups = env.eng.t['lookups']
    lookups.zero_()
    if args.policy == 'pure_pursuit':
        # secondary mode: the reference's waypoint follower runs on the GPU in front of every step
        rl = workload.load_waypoints(workload.RACELINE)
        wp = torch.as_tensor(np.ascontiguousarray(rl[:, [1, 2, 5]]), device=dev)
        step_fn = lambda k: env.step(env.pure_pursuit(wp, 0.82461887897713965, 1.375))  # noqa: E731
        for k in range(W):
            step_fn(k)
        lookups.zero_()
    else:
        step_fn = lambda k: env.step(acts[(W + k) % POOL])  # noqa: E731
    if args.bitmap != 'none':
        from red_gym_amd.lidar import LidarBitmap
        to_img = LidarBitmap(1080, bg_color='black', draw_mode=args.bitmap, device=local_rank)
        imgs = torch.empty((B, 256, 256), dtype=torch.uint8, device=dev)
        inner = step_fn

        def step_fn(k):  # noqa: F811
            obs = inner(k)[0]
            to_img(obs['scans'][:, 0], out=imgs)
        step_fn(0)
        lookups.zero_()
    NEVER = 1 << 30  # a sampling period no run reaches: the measurement aid active, no step instrumented
    if not args.no_scan_events:
        env.eng.profile_begin(K, every=max(1, args.scan_events_every))  # hipEvent pairs + lookup counting on every Nth timed scan launch
    else:
        env.eng.profile_begin(1, every=NEVER)
    elapsed = timed_steps(ranks, step_fn, K)
    per_rank = list(ranks.last_per_rank)                                      # every rank's own elapsed seconds
    scan_prof = env.eng.profile_end()                                         # the sampled launches of the K timed steps only
    if args.no_scan_events:
        scan_prof = None
    tot_lookups = int(lookups.to(torch.int64).sum().item())                  # ... and their table reads
    repeats = None
    if args.repeats > 1:
        env.eng.profile_begin(1, every=NEVER)
        ms = [elapsed / K * 1e3]
        for r in range(1, args.repeats):
            ms.append(timed_steps(ranks, lambda k, r=r: step_fn(r * K + k), K) / K * 1e3)
        env.eng.profile_end()
        med = float(np.median(ms))
        repeats = {'ms_per_step': ms, 'median_ms_per_step': med, 'value': world * B / (med * 1e-3), 'unit': 'env-steps/s',
                   'note': 'region 0 is the protocol region (`value`); regions 1.. follow it back to back, uninstrumented'}
    names = ranks.gather_names(torch.cuda.get_device_name(dev))
    sustained = None
    if args.sustained > 0:
        env.eng.profile_begin(1, every=NEVER)                                # uninstrumented, like the un-sampled timed steps
        el2 = timed_steps(ranks, lambda k: step_fn(K + k), args.sustained)
        env.eng.profile_end()
        sustained = {'value': world * B * args.sustained / el2, 'unit': 'env-steps/s', 'steps': args.sustained,
                     'ms_per_step': el2 / args.sustained * 1e3,
                     'note': 'the %d steps right after the timed region (GPU clocks up, cars scattered by random '
                             'driving); not part of `value`' % args.sustained}

    steady = None
    if args.steady_state > 0 and env.eng._noise_on:
        # `value`, `sustained` and the regions above run right after a reset: every car stands on the SAME noise row, which
        # the whole chip then reads from the L1.  A batch that has been running for a while (autoreset at different
        # times) has every env on its own row, and the rows stream from L2 / HBM.  That regime is set up directly: the
        # envs' row counters are staggered over 1 024 rows (a device-side write; the host's upper bound follows), a few
        # steps settle it, and a further region is timed.
        ns = env.eng.t['noise_step']
        stagger = torch.randint(1, 1025, (B, 1), device=dev, dtype=ns.dtype, generator=torch.Generator(device=dev).manual_seed(5 + rank))
        ns.copy_(ns + stagger)
        env.eng.host_steps_bound += 1024
        for k in range(20):
            step_fn(k)
        env.eng.profile_begin(1, every=NEVER)
        el3 = timed_steps(ranks, lambda k: step_fn(K + k), args.steady_state)
        env.eng.profile_end()
        steady = {'value': world * B * args.steady_state / el3, 'unit': 'env-steps/s', 'steps': args.steady_state,
                  'ms_per_step': el3 / args.steady_state * 1e3,
                  'note': 'every env on its own noise row (row counters staggered over 1 024 rows, as in a batch whose envs were '
                          'reset at different times): the rows stream from L2 / HBM instead of one row from the L1; `value` is '
                          'the protocol region right after a common reset, the friendliest regime for the noise gather'}

    out = None
    if rank == 0:
        value = world * B * K / elapsed
        roof = None
        if not args.no_scan_events:
            scan_ms, n_launch = scan_prof
            cars = B * A
            # SURVEY 8(d) byte model, per car-step: L*4 + 1080*4 + 72
            # lookups are counted by the launches that carry the events (f110_profile_every), so bytes and time are
            # those of the same n_launch dispatches
            bytes_per_launch = (tot_lookups / max(n_launch, 1)) * 4.0 + cars * (1080 * 4 + 72)
            avg_s = scan_ms * 1e-3 / max(n_launch, 1)
            achieved = bytes_per_launch / avg_s / 1e9
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate
            # runs of this same workload; tools/profile.sh writes profiles/traffic.json)
            traffic = None
            tp = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tp):
                tj = json.load(open(tp))
                key = '%dx%d' % (B, A)
                if key in tj:
                    traffic = tj[key]['hbm_bytes_per_launch']
            # `achieved` / `frac` are SURVEY 8(d)'s MODEL: algorithmic bytes (4 B per distance-table lookup + the fp32
            # scan + state) over the kernel time, priced against HBM peak as the survey prescribes.  What the
            # counters say is spelled out beside it: the lookups are served by L1 / L2, real HBM traffic is the
            # scan write (traffic_*), and the kernel is bound by VALU issue and L1->L2 gather traffic.
            roof = {'bound': 'hbm', 'kernel': 'scan_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                    'traffic_source': 'profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER run of '
                                      'this workload, tools/profile.sh; not collected by this run)' if traffic else None,
                    'avg_launch_ms': avg_s * 1e3, 'launches': n_launch,
                    'lookups_per_car_step': tot_lookups / max(n_launch, 1) / cars, 'events_every': max(1, args.scan_events_every),
                    'algorithmic_bytes_per_launch': bytes_per_launch,
                    'model': 'SURVEY 8(d) algorithmic bytes: lookups*4 + cars*(1080*4 + 72), not HBM traffic',
                    'traffic_gbs': (traffic / avg_s / 1e9) if traffic else None,
                    'traffic_frac': (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    'limiter': 'the L1 address / tag path (368 M accesses + 90 M misses per 65 536-car launch: every look-up of a '
                               'marching ray is one) with VALU issue at ~72 % of the SIMD cycles and the latency of the refill '
                               'phases; not HBM; see DESIGN.md 5 and profiles/r05_scan_budget.txt'}
        out = {'metric': 'env steps/sec (all envs), 1080-beam lidar', 'value': value, 'unit': 'env-steps/s',
               'n_gpus': world, 'steps': K, 'warmup': W, 'spinup_steps': args.spinup, 'ms_per_step': elapsed / K * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
               'data': 'synthetic',
               'config': {'workload': '%d envs x %d agent(s) per GPU, example_map, 1080 beams, RK4 single-track, '
                                      'noise+iTTC%s, lap logic, autoreset; %s'
                                      % (B, A, '+GJK+opponent ray-cast' if A > 1 else '',
                                         ('random actions' if args.policy == 'random' else 'GPU pure-pursuit policy in the loop')
                                         + ('' if args.bitmap == 'none' else '; + %s bitmap of every ego scan' % args.bitmap)),
                          'envs_per_gpu': B, 'agents': A, 'num_beams': 1080, 'map': 'example_map',
                          'sharding': 'independent env shards, no collective on the step path'},
               'per_rank_ms': [t / K * 1e3 for t in per_rank], 'devices': names,
               'rank_sync': ('rccl' if ranks.backend == 'nccl' else ranks.backend) if ranks.grouped else 'none (one rank)',
               'roofline': roof, 'sustained': sustained, 'steady_state': steady}
        if repeats:
            out['median_of_repeats'] = repeats
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(A)
    env.close()
    ranks.close()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    sys.exit(main() or 0)
