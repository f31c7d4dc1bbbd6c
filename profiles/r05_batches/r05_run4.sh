cd $GRAFT_REPO_ROOT
F110_LIB=variants_ship/timeline.so timeout -k 10 200 python tools/timeline.py --envs 65536 > gpurun_out/r05_timeline_65536.txt 2>&1 || { tail -5 gpurun_out/r05_timeline_65536.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05_timeline_65536.txt | head -30
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_b.txt 2>&1; rc=$?
tail -4 gpurun_out/r05_gputest_b.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py > gpurun_out/r05_bench_default_a.json 2> gpurun_out/r05_bench_default_a.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_a.json 2> gpurun_out/r05_bench_driver_a.err || exit 1
python - <<'PY'
import json
for f in ('gpurun_out/r05_bench_default_a.json','gpurun_out/r05_bench_driver_a.json'):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j['roofline']
    print(f, 'value %.2f M  ms/step %.4f  scan %.4f ms  frac %.3f  sustained %s' % (j['value']/1e6, j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('sustained')))
PY
