#!/bin/bash
# Bench lines of every single-GPU configuration into <outdir> (default gpurun_out/bench_all): one gpurun call.
set -u
O=${1:-gpurun_out/bench_all}; mkdir -p $O
T="timeout -k 10 500"
$T python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err
$T python bench.py > $O/bench_default.json 2> $O/bench_default.err
$T python bench.py --envs 4096 --no-cpu-baseline > $O/bench_4096x1.json 2>/dev/null
$T python bench.py --envs 16384 --agents 2 --no-cpu-baseline > $O/bench_16384x2.json 2>/dev/null
$T python bench.py --policy pure_pursuit --no-cpu-baseline > $O/bench_pure_pursuit.json 2>/dev/null
$T python bench.py --bitmap FILL --no-cpu-baseline > $O/bench_with_bitmap.json 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); r=d.get('roofline') or {}
print('%-28s value %.2f M  ms/step %.4f  scan %.4f ms (n=%s) frac %.3f  sustained %.2f M' % ('$f'.split('/')[-1], d['value']/1e6, d['ms_per_step'], r.get('avg_launch_ms') or 0, r.get('launches'), r.get('frac') or 0, (d.get('sustained') or {}).get('value',0)/1e6))"; done
