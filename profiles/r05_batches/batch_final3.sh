#!/bin/bash
set -e
mkdir -p gpurun_out/final3
timeout -k 10 600 python -m pytest tests/test_gpu_bitmap.py -x -q 2>&1 | tail -1
timeout -k 10 300 python bench.py --bitmap FILL --no-cpu-baseline > gpurun_out/final3/bench_with_bitmap.json 2> gpurun_out/final3/bench_with_bitmap.err
python3 -c "
import json; d=json.load(open('gpurun_out/final3/bench_with_bitmap.json')); print('with bitmap: %.2f M env-steps/s, %.4f ms/step' % (d['value']/1e6, d['ms_per_step']))"
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/final3/prof_bitmap -o bm -- python3 $ROOT/tools/bench_bitmap.py --reps 20 > $ROOT/gpurun_out/final3/prof_bitmap.txt 2> $ROOT/gpurun_out/final3/prof_bitmap.err
cd $ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/final3/prof_bitmap/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bitmap' in r['Name'] or 'occupancy' in r['Name']:
        print(r['Name'][:60], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1))
PY
