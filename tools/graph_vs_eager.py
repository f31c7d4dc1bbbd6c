"""Eager launches vs HIP-graph replays of the step:
    python tools/graph_vs_eager.py [mode ...] [size ...]
modes: eager  torch (torch.cuda.graph capture)  nodes (library-built graph of explicit kernel nodes)  capture (library
capture on a private non-blocking stream)  torch_nb (torch capture, replayed on a non-blocking side stream)
Also writes the dot dump of the library graph and of the torch graph under gpurun_out/graphs/ when that directory
exists.  Run one mode under `rocprofv3 --kernel-trace` and feed the trace to tools/trace_gaps.py for kernel durations
and the idle gaps between consecutive dispatches."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from red_gym_amd import F110VecEnv, workload
modes = [m for m in sys.argv[1:] if not m.isdigit()] or ['eager', 'torch', 'nodes', 'capture', 'eager']
sizes = [int(m) for m in sys.argv[1:] if m.isdigit()] or [65536, 4096]
N = 200
dump = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'graphs')
for B in sizes:
    res = []
    for mode in modes:
        env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
        env.reset(torch.as_tensor(workload.spawn_poses(B, 1), device=env.device))
        acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
        for k in range(60): env.step(acts[k % 8])
        if mode == 'eager':
            fn = lambda k: env.step(acts[k % 8])
        elif mode in ('torch', 'torch_nb'):
            buf = env.capture_step()
            buf.copy_(acts[0])
            if os.path.isdir(dump):
                try:
                    g = torch.cuda.CUDAGraph(); g.enable_debug_mode()
                    side = torch.cuda.Stream()
                    with torch.cuda.stream(side):
                        with torch.cuda.graph(g, stream=side):
                            env.eng.step(buf)
                    env.eng.host_steps_bound -= 1
                    g.debug_dump(os.path.join(dump, 'torch_%d.dot' % B))
                except Exception as e:  # diagnostics only
                    print('torch dot dump failed:', e)
            fn = lambda k: env.step_graph()
        else:
            buf = env.build_step_graph(mode)
            buf.copy_(acts[0])
            n = env.lib_graph_info(os.path.join(dump, 'lib_%s_%d.dot' % (mode, B)) if os.path.isdir(dump) else None)
            fn = lambda k: env.step_lib_graph()
        for k in range(20): fn(k)
        torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        for k in range(N):
            h0 = time.perf_counter(); fn(k); host += time.perf_counter() - h0
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
        res.append('%s %.4f ms (host call %.1f us)' % (mode, dt * 1e3, host / N * 1e6))
        env.close()
    print(B, '  '.join(res), flush=True)
