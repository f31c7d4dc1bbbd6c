cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_rccl.py -x -q > gpurun_out/r05_fullsize.txt 2>&1; rc=$?
tail -8 gpurun_out/r05_fullsize.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r05_bench_default_b.json 2> gpurun_out/r05_bench_default_b.err || { tail -5 gpurun_out/r05_bench_default_b.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05_bench_default_b.json').read().strip().splitlines()[-1]); r=j['roofline']
print('value %.2f M  ms/step %.4f  scan %.4f ms  frac %.3f' % (j['value']/1e6, j['ms_per_step'], r['avg_launch_ms'], r['frac']))
print('sustained', j['sustained']['value']/1e6, 'steady_state', j['steady_state']['value']/1e6, j['steady_state']['ms_per_step'])
print('cpu', j['cpu_baseline'])
PY
