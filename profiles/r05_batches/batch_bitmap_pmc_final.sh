#!/bin/bash
ROOT=$PWD; mkdir -p $ROOT/gpurun_out/bmpmc; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $ROOT/gpurun_out/bmpmc/a -o pmc -- python3 $ROOT/tools/bench_bitmap.py --reps 3 > $ROOT/gpurun_out/bmpmc/a.out 2> $ROOT/gpurun_out/bmpmc/a.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/bmpmc/w -o pmc -- python3 $ROOT/tools/bench_bitmap.py --reps 3 > $ROOT/gpurun_out/bmpmc/w.out 2> $ROOT/gpurun_out/bmpmc/w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/bmpmc/f -o pmc -- python3 $ROOT/tools/bench_bitmap.py --reps 3 > $ROOT/gpurun_out/bmpmc/f.out 2> $ROOT/gpurun_out/bmpmc/f.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ('a', 'w', 'f'):
    f = glob.glob('gpurun_out/bmpmc/%s/**/*counter_collection.csv' % d, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if 'bitmap_kernel' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    print({k: round(sum(x)/len(x)/1e6, 2) for k, x in acc.items()}, flush=True)
PY
