"""Row (e) on real hardware, as far as ONE GPU allows: the collective calls bench.py makes around its timed region
(RCCL init with device_id, barrier(device_ids), all_gather, all_reduce(MAX), destroy_process_group) executed by a forced
ONE-rank RCCL process group in a fresh child process, around real f110_step launches; and two handles driven from one
process on two streams."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_a_forced_one_rank_rccl_group():
    """`bench.py` with F110_BENCH_FORCE_GROUP=1: backend 'nccl' (= RCCL), world size 1, 4 096 envs.  The group is built
    in a fresh process before any GPU call, exactly as a rank of the multi-GPU run builds it."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'F110_BENCH_BACKEND')}
    env.update(F110_BENCH_FORCE_GROUP='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--envs', '4096', '--steps', '10',
                        '--warmup', '3', '--sustained', '0', '--no-cpu-baseline', '--repeats', '3'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['steps'] == 10 and out['data'] == 'synthetic'
    assert len(out['per_rank_ms']) == 1 and 0 < out['per_rank_ms'][0] <= out['ms_per_step']
    assert len(out['devices']) == 1 and out['devices'][0]           # the device name came back through all_gather
    assert out['value'] > 1e6 and out['roofline']['launches'] >= 2
    assert len(out['median_of_repeats']['ms_per_step']) == 3
    assert out['rank_sync'] == 'rccl', p.stderr[-2000:]        # RCCL itself ran (bench.py falls back to gloo only if it cannot start)


def test_bench_as_a_torchrun_rank_with_rccl():
    """The driver's N > 1 form -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 1 and the forced group: RANK / WORLD_SIZE / MASTER_* come from the
    launcher, the rendezvous is the launcher's, the collectives are RCCL's."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'F110_BENCH_BACKEND')}
    env.update(F110_BENCH_FORCE_GROUP='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--envs', '2048', '--steps', '5',
                        '--warmup', '2', '--sustained', '0', '--no-cpu-baseline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['steps'] == 5 and len(out['per_rank_ms']) == 1 and len(out['devices']) == 1
    assert out['rank_sync'] == 'rccl', p.stderr[-2000:]


def test_two_handles_interleaved_on_two_streams(assets):
    """Two F110VecEnv handles in one process, stepped alternately on two streams, give what each gives alone: a handle
    restores the caller's current device after every allocating call and launches only on the caller's stream."""
    import torch
    from red_gym_amd import F110VecEnv, workload
    B, A, T = 64, 2, 25
    mp = os.path.join(assets, 'example_map')

    def run_alone(seed_off):
        e = F110VecEnv(B, map=mp, num_agents=A, autoreset=True, keep_f64_scans=True)
        poses = workload.spawn_poses(B, A, seed_off)
        acts = torch.as_tensor(workload.action_pool(8, B, A, seed_off), device='cuda')
        e.reset(poses)
        for k in range(T):
            e.step(acts[k % 8])
        torch.cuda.synchronize()
        res = {k: v.clone() for k, v in e.eng.t.items() if v is not None}
        e.close()
        return res

    ref = [run_alone(0), run_alone(1)]
    cur = torch.cuda.current_device()
    envs = [F110VecEnv(B, map=mp, num_agents=A, autoreset=True, keep_f64_scans=True) for _ in range(2)]
    assert torch.cuda.current_device() == cur
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    acts = [torch.as_tensor(workload.action_pool(8, B, A, i), device='cuda') for i in range(2)]
    torch.cuda.synchronize()
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            envs[i].reset(workload.spawn_poses(B, A, i))
    for k in range(T):
        for i in (0, 1) if k % 2 == 0 else (1, 0):
            with torch.cuda.stream(streams[i]):
                envs[i].step(acts[i][k % 8])
    torch.cuda.synchronize()
    for i in range(2):
        for key, v in ref[i].items():
            assert torch.equal(envs[i].eng.t[key], v), (i, key)
        envs[i].close()


def test_handle_refuses_a_foreign_current_device(assets):
    """f110_step with another device current: F110_E_INVALID naming both devices (needs two GPUs)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('one visible GPU')
    import ctypes as C
    from red_gym_amd import F110VecEnv, _lib, workload
    e = F110VecEnv(8, map=os.path.join(assets, 'example_map'), num_agents=1, device=0)
    e.reset(workload.spawn_poses(8, 1))
    a = torch.zeros((8, 1, 2), dtype=torch.float64, device='cuda:0')
    with torch.cuda.device(1):
        e.step(a)                                     # the engine makes its device current itself
        assert torch.cuda.current_device() == 1
        rc = e.eng.lib.f110_step(e.eng._h, C.c_void_p(a.data_ptr()), None)   # the raw ABI does not: refused
        assert rc == _lib.E_INVALID and b'current device is 1' in e.eng.lib.f110_last_error()
        e2 = F110VecEnv(8, map=os.path.join(assets, 'example_map'), num_agents=1, device=0)   # allocating calls restore
        assert torch.cuda.current_device() == 1
        e2.close()
    e.close()
