"""The N>1 path of bench.py on CPU: two gloo ranks, each owning an independent env shard
(no collective on the step path); barriers around the timed region, MAX over ranks."""
import os
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import bench
    ranks = bench.Ranks('gloo', None)
    poses, acts = bench.rank_workload(ranks.rank, 64, 2, pool=4)
    steps = []

    def step(k):  # stand-in for env.step: rank 1 is the slow shard
        steps.append(k)
        time.sleep(0.01 * (1 + ranks.rank))
    elapsed = bench.timed_steps(ranks, step, 5)
    names = ranks.gather_names('dev-of-rank-%d' % rank)
    q.put((rank, elapsed, len(steps), float(poses.sum()), float(acts.sum()), list(ranks.last_per_rank), names))
    ranks.close()


def test_two_rank_sharding_and_timing():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, n0, ps0, as0, pr0, nm0), (r1, e1, n1, ps1, as1, pr1, nm1) = res
    assert pr0 == pr1 and len(pr0) == 2 and max(pr0) <= e0 and pr0[1] > pr0[0] * 1.5   # every rank's own time: the straggler shows
    assert nm0 == nm1 == ['dev-of-rank-0', 'dev-of-rank-1']
    assert (r0, r1) == (0, 1) and n0 == n1 == 5          # EXACTLY K steps on every rank
    assert e0 == e1                                      # MAX over ranks is what both report
    assert e0 >= 5 * 0.02 * 0.9                          # the slow rank (20 ms/step) sets it
    assert ps0 != ps1 and as0 != as1                     # independent shards (different seeds)


def _worker_vote(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), F110_BENCH_TEST_PRIMARY_FAILS='1')
    import bench
    ranks = bench.Ranks('gloo', None)           # the primary backend "fails" on rank 1 only
    e = bench.timed_steps(ranks, lambda k: time.sleep(0.002), 3)
    q.put((rank, ranks.backend, ranks.dist.get_world_size(), e, ranks.gather_names('r%d' % rank)))
    ranks.close()


def test_backend_fallback_is_decided_by_all_ranks_together():
    """a rank whose RCCL probe fails must take EVERY rank to the fallback group -- also the ranks whose probe
    succeeded (they used to stay in the RCCL group, and the first barrier never returned).  The primary backend is made to
    fail on rank 1 only; both ranks must end up in the same (fallback) group and run the timed region together."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_vote, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1] and all(r[1] == 'gloo' and r[2] == 2 for r in res)
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4] == ['r0', 'r1']


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
    import bench
    ranks = bench.Ranks('gloo', None)
    assert ranks.world == 1 and ranks.max_over_ranks(1.5) == 1.5
    n = []
    assert bench.timed_steps(ranks, lambda k: n.append(k), 7) >= 0 and n == list(range(7))
    assert ranks.last_per_rank is not None and len(ranks.last_per_rank) == 1 and ranks.gather_names('x') == ['x']
    ranks.close()


def test_forced_one_rank_group_runs_the_collectives():
    """force_group=True builds a process group for ONE rank (what tests/test_gpu_rccl.py does with RCCL on the GPU box):
    barrier, all_gather, all_reduce(MAX) and destroy_process_group execute."""
    import subprocess
    code = ('import sys; sys.path.insert(0, %r); import bench; r = bench.Ranks("gloo", None, force_group=True); '
            'assert r.grouped and r.dist.is_initialized() and r.dist.get_world_size() == 1; '
            'e = bench.timed_steps(r, lambda k: None, 3); assert len(r.last_per_rank) == 1 and r.last_per_rank[0] <= e; '
            'assert r.gather_names("abc") == ["abc"]; r.close(); assert not r.dist.is_initialized(); print("ok")' % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith('ok'), p.stderr


def _run_bench(args, extra_env, timeout=180):
    import json
    import subprocess
    env = dict(os.environ, F110_BENCH_BACKEND='gloo', **extra_env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_self_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command form): the parent starts two fresh
    rank processes, rank 0 prints ONE JSON line whose n_gpus is the process group's world size, value aggregates
    both shards and the slow rank (2x the step time) sets ms_per_step."""
    rc, lines, err = _run_bench(['--gpus', '2', '--steps', '6', '--warmup', '2', '--envs', '32'],
                                {'F110_BENCH_STUB_STEP_MS': '10'})
    assert rc == 0, err
    assert len(lines) == 1
    out = lines[0]
    assert out['n_gpus'] == 2 and out['steps'] == 6 and out['warmup'] == 2 and out['data'] == 'stub'
    assert out['ms_per_step'] >= 20 * 0.9                       # rank 1 sleeps 20 ms per step
    assert len(out['per_rank_ms']) == 2 and out['per_rank_ms'][1] > 1.5 * out['per_rank_ms'][0]
    assert max(out['per_rank_ms']) <= out['ms_per_step'] and out['devices'] == ['stub-cpu-0', 'stub-cpu-1']
    assert abs(out['value'] - 2 * 32 * 6 / (out['ms_per_step'] * 6e-3)) < 1e-6 * out['value']


def test_ranks_started_by_torch_distributed_run():
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...`.  The launcher's agent hosts the rendezvous store on that port, so the ranks must
    join it as clients (rank 0 binding the port itself fails with "address already in use")."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, F110_BENCH_BACKEND='gloo', F110_BENCH_STUB_STEP_MS='5')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--envs', '16'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['steps'] == 4 and len(lines[0]['per_rank_ms']) == 2


def test_self_launch_propagates_a_rank_failure():
    """A rank that dies must not leave the others waiting at a barrier: non-zero exit code, no JSON line."""
    rc, lines, err = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '0', '--envs', '8'],
                                {'F110_BENCH_STUB_STEP_MS': 'not-a-number'})
    assert rc != 0 and lines == []


def test_launcher_never_touches_torch_in_the_parent():
    """The self-launch must happen before torch is imported (a process that initialised the GPU may not start
    the ranks by re-exec, and must not hold the device the children need)."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main_src = src[src.index('def main('):]
    assert main_src.index('self_launch(args.gpus, argv)') < main_src.index('import torch')
    launch_src = src[src.index('def self_launch('):src.index('def stub_bench(')]
    assert 'import torch' not in launch_src and 'os.exec' not in launch_src and 'subprocess.Popen' in launch_src
