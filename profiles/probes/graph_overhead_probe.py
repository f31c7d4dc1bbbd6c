"""Is the per-replay cost of a captured step ours or the runtime's?  Three plain torch kernels that write N bytes each,
eager vs one torch.cuda.CUDAGraph replay, for growing N (the step at 65 536 envs writes 283 MB of scans):
    python tools/graph_overhead_probe.py"""
import time
import torch

dev = torch.device('cuda', 0)
for mb in (1, 16, 64, 283, 1024):
    n = mb * (1 << 20) // 4
    a, b, c = (torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3))

    def step():
        a.add_(1.0); b.add_(1.0); c.add_(1.0)
    for _ in range(20):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.current_stream(dev).wait_stream(s)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 200
    print('%5d MB per kernel x 3: eager %.4f ms  graph %.4f ms  (graph - eager = %+.1f us)' % (mb, eager * 1e3, graph * 1e3, (graph - eager) * 1e6), flush=True)
