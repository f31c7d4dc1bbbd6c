#!/bin/bash
# PMC passes of the bitmap kernel: old vs new variants (FILL, 65 536 images)
mkdir -p $PWD/gpurun_out/bm4
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
for v in oldbm tree s1 d4; do
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$ROOT/variants_ship/$v.so F110_LIB_OLDER=1; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $ROOT/gpurun_out/bm4/pmc_$v -o pmc -- python3 $ROOT/tools/bench_bitmap.py --reps 3 > $ROOT/gpurun_out/bm4/$v.out 2> $ROOT/gpurun_out/bm4/$v.err || exit 1
  echo "$v done"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for v in ('oldbm','tree','s1','d4'):
    f = glob.glob('gpurun_out/bm4/pmc_%s/**/*counter_collection.csv' % v, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if 'bitmap_kernel' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    print(v, {k: round(sum(x)/len(x)/1e6, 2) for k, x in acc.items()}, flush=True)
PY
