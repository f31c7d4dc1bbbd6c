"""Eager launches vs hipGraph replay of the step, with the pieces that could explain a gap separated:
    python tools/graph_vs_eager.py [mode ...]     modes: eager graph_copy graph_nocopy   (default: all)
Run one mode under `rocprofv3 --kernel-trace` and feed the trace to tools/trace_gaps.py to see kernel
durations and the idle gaps between consecutive dispatches."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from red_gym_amd import F110VecEnv, workload
modes = [m for m in sys.argv[1:] if not m.isdigit()] or ['eager', 'graph_copy', 'graph_nocopy', 'graph_1exec']
sizes = [int(m) for m in sys.argv[1:] if m.isdigit()] or [65536, 4096]
N = 200
for B in sizes:
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
    env.reset(torch.as_tensor(workload.spawn_poses(B, 1), device=env.device))
    acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
    for k in range(30): env.step(acts[k % 8])
    res = {}
    if 'eager' in modes:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(N): env.step(acts[k % 8])
        torch.cuda.synchronize(); res['eager'] = (time.perf_counter() - t0) / N
    if 'graph_copy' in modes or 'graph_nocopy' in modes:
        buf = env.capture_step(copies=2)
        for k in range(10): env.step_graph(acts[k % 8])
    if 'graph_copy' in modes:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(N): env.step_graph(acts[k % 8])
        torch.cuda.synchronize(); res['graph_copy'] = (time.perf_counter() - t0) / N
    if 'graph_nocopy' in modes:       # actions already sit in the static buffer
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(N): env.step_graph()
        torch.cuda.synchronize(); res['graph_nocopy'] = (time.perf_counter() - t0) / N
    if 'graph_1exec' in modes:        # ONE graph exec replayed back to back (what round 1 measured)
        env.capture_step(copies=1)
        for k in range(10): env.step_graph()
        torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        for k in range(N):
            h0 = time.perf_counter(); env.step_graph(); host += time.perf_counter() - h0
        torch.cuda.synchronize(); res['graph_1exec'] = (time.perf_counter() - t0) / N
        res['graph_1exec_host_call'] = host / N
    print(B, '  '.join('%s %.4f ms' % (k, v * 1e3) for k, v in res.items()), flush=True)
    env.close()
