cd $GRAFT_REPO_ROOT
bash tools/profile_default.sh > gpurun_out/r05_default_trace.txt 2>&1; cat gpurun_out/r05_default_trace.txt
bash tools/bench_all.sh gpurun_out/bench_r05 2>&1 | tail -8
O=gpurun_out/bench_r05
T="timeout -k 10 500"
$T python bench.py --steps 1000 --warmup 200 --repeats 5 > $O/protocol_65536x1.json 2> $O/protocol_65536x1.err
$T python bench.py --steps 1000 --warmup 200 --repeats 5 --envs 4096 > $O/protocol_4096x1.json 2> $O/protocol_4096x1.err
$T python bench.py --steps 1000 --warmup 200 --repeats 5 --envs 16384 --agents 2 > $O/protocol_16384x2.json 2> $O/protocol_16384x2.err
for f in $O/protocol_*.json; do python3 -c "
import json; d=json.load(open('$f')); r=d['roofline']; m=d['median_of_repeats']; c=d['cpu_baseline']
print('%-28s value %.2f M  median-of-5 %.2f M (%.4f ms)  scan %.4f ms frac %.3f  steady %.2f M  cpu %.0f (%d thr) / %.0f (1 thr)' % ('$f'.split('/')[-1], d['value']/1e6, m['value']/1e6, m['median_ms_per_step'], r['avg_launch_ms'], r['frac'], d['steady_state']['value']/1e6, c['value'], c['cores'], c['single_thread_value']))"; done
