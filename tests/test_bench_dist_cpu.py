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
    q.put((rank, elapsed, len(steps), float(poses.sum()), float(acts.sum())))
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
    (r0, e0, n0, ps0, as0), (r1, e1, n1, ps1, as1) = res
    assert (r0, r1) == (0, 1) and n0 == n1 == 5          # EXACTLY K steps on every rank
    assert e0 == e1                                      # MAX over ranks is what both report
    assert e0 >= 5 * 0.02 * 0.9                          # the slow rank (20 ms/step) sets it
    assert ps0 != ps1 and as0 != as1                     # independent shards (different seeds)


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
    import bench
    ranks = bench.Ranks('gloo', None)
    assert ranks.world == 1 and ranks.max_over_ranks(1.5) == 1.5
    n = []
    assert bench.timed_steps(ranks, lambda k: n.append(k), 7) >= 0 and n == list(range(7))
    ranks.close()
