cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_mirrors.py -x -q > gpurun_out/r05_mirrors.txt 2>&1; rc=$?
tail -6 gpurun_out/r05_mirrors.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/time_planner.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_planner.txt
timeout -k 10 300 python bench.py --policy pure_pursuit --no-cpu-baseline --steady-state 0 > gpurun_out/r05_bench_pp.json 2>/dev/null
python -c "
import json; j=json.loads(open('gpurun_out/r05_bench_pp.json').read().strip().splitlines()[-1]); print('pure_pursuit policy: %.2f M env-steps/s, %.4f ms/step' % (j['value']/1e6, j['ms_per_step']))" | tee -a gpurun_out/r05_planner.txt
