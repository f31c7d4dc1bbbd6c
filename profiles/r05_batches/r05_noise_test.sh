cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_noise.py -x -q > gpurun_out/r05_noise_test.txt 2>&1; rc=$?
tail -25 gpurun_out/r05_noise_test.txt
[ $rc -eq 0 ] || exit $rc
cat > /tmp/per_env_bench.py <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from red_gym_amd import F110VecEnv, workload
B = 65536
for src, seed in (('device', 12345), ('per_env', list(range(B)))):
    t0 = time.perf_counter()
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True, seed=seed, noise_source=src)
    t1 = time.perf_counter()
    poses = torch.as_tensor(workload.spawn_poses(B, 1), device=env.device)
    acts = torch.as_tensor(workload.action_pool(8, B, 1), device=env.device)
    env.reset(poses)
    for k in range(100): env.step(acts[k % 8])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for k in range(200): env.step(acts[k % 8])
    torch.cuda.synchronize(); dt = time.perf_counter() - t2
    print('%-8s 65536 envs: %.3f ms/step, %.2f M env-steps/s (set-up %.2f s)' % (src, dt / 200 * 1e3, B * 200 / dt / 1e6, t1 - t0), flush=True)
    env.close()
PY
timeout -k 10 300 python /tmp/per_env_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_per_env_cost.txt
