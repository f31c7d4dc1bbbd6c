import torch, time
torch.cuda.synchronize()
for c in (1e8, 1e9):
    t0=time.perf_counter(); torch.cuda._sleep(int(c)); torch.cuda.synchronize(); print(c, time.perf_counter()-t0)
