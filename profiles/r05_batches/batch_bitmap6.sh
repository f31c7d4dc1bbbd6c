#!/bin/bash
set -e
mkdir -p gpurun_out/bm6
timeout -k 10 600 python -m pytest tests/test_gpu_bitmap.py -x -q > gpurun_out/bm6/tests.log 2>&1 || { tail -30 gpurun_out/bm6/tests.log; exit 1; }
tail -1 gpurun_out/bm6/tests.log
for rep in 1 2 3; do
for v in tree oldbm; do
  echo "## $v" | tee -a gpurun_out/bm6/bench.log
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  timeout -k 10 200 python tools/bench_bitmap.py --mode FILL --reps 100 2>&1 | grep "^bitmap" | tee -a gpurun_out/bm6/bench.log
done
done
unset F110_LIB F110_LIB_OLDER
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $ROOT/gpurun_out/bm6/pmc_tree -o pmc -- python3 $ROOT/tools/bench_bitmap.py --reps 3 > $ROOT/gpurun_out/bm6/pmc.out 2> $ROOT/gpurun_out/bm6/pmc.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/bm6/pmc_tree/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if 'bitmap_kernel' in row['Kernel_Name']:
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
print('tree', {k: round(sum(x)/len(x)/1e6, 2) for k, x in acc.items()}, flush=True)
PY
