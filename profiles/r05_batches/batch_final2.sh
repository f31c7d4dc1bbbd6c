#!/bin/bash
# final refresh: bench lines of every configuration, the default run's kernel trace, the bitmap's kernel stats
set -e
bash tools/bench_all.sh gpurun_out/bench_r05c | tee gpurun_out/bench_r05c.txt
bash tools/profile_default.sh | tee gpurun_out/prof_r05c_default.txt
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r05c_bitmap -o bm -- python3 $ROOT/tools/bench_bitmap.py --reps 20 > $ROOT/gpurun_out/prof_r05c_bitmap.txt 2> $ROOT/gpurun_out/prof_r05c_bitmap.err
cd $ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r05c_bitmap/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bitmap' in r['Name'] or 'occupancy' in r['Name']:
        print(r['Name'][:60], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1))
PY
