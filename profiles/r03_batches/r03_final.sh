#!/bin/bash
# round-3 bench lines and profiles of the committed state (one gpurun call)
set -u
O=gpurun_out/r03z; mkdir -p $O
T="timeout -k 10 500"
for m in "FILL 1" "POLYGON 1"; do set -- $m; echo -n "bm1024 " >> $O/bitmap_1024.txt; F110_LIB=$PWD/variants_ship/bm1024.so $T python tools/bench_bitmap.py --mode $1 --channels $2 2>&1 | grep "^bitmap" >> $O/bitmap_1024.txt; echo -n "bm512  " >> $O/bitmap_1024.txt; $T python tools/bench_bitmap.py --mode $1 --channels $2 2>&1 | grep "^bitmap" >> $O/bitmap_1024.txt; done
cat $O/bitmap_1024.txt
$T python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err
$T python bench.py > $O/bench_default.json 2> $O/bench_default.err
$T python bench.py --envs 4096 --no-cpu-baseline > $O/bench_4096x1.json 2>/dev/null
$T python bench.py --envs 16384 --agents 2 --no-cpu-baseline > $O/bench_16384x2.json 2>/dev/null
$T python bench.py --policy pure_pursuit --no-cpu-baseline > $O/bench_pure_pursuit.json 2>/dev/null
$T python bench.py --bitmap FILL --no-cpu-baseline > $O/bench_with_bitmap.json 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); r=d.get('roofline') or {}
print('%-28s value %.2f M  ms/step %.4f  scan %.4f ms frac %.3f  sustained %.2f M' % ('$f'.split('/')[-1], d['value']/1e6, d['ms_per_step'], r.get('avg_launch_ms') or 0, r.get('frac') or 0, (d.get('sustained') or {}).get('value',0)/1e6))"; done
BITMAP=1 timeout -k 10 900 bash tools/profile.sh r03_65536x1 > $O/profile.log 2>&1; tail -40 $O/profile.log
