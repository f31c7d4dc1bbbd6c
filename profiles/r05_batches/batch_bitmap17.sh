#!/bin/bash
set -e
mkdir -p gpurun_out/bm20
timeout -k 10 600 python -m pytest tests/test_gpu_bitmap.py -x -q > gpurun_out/bm20/tests.log 2>&1 || { tail -30 gpurun_out/bm20/tests.log; exit 1; }
tail -1 gpurun_out/bm20/tests.log
for rep in 1 2; do
for args in "--mode FILL" "--mode POLYGON" "--mode RAYS" "--mode FILL --channels 3" "--mode FILL --channels 4" "--mode FILL --beams 1079"; do
for v in tree oldbm; do
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  echo -n "$v: " | tee -a gpurun_out/bm20/bench.log
  timeout -k 10 200 python tools/bench_bitmap.py $args --reps 50 2>&1 | grep "^bitmap" | tee -a gpurun_out/bm20/bench.log
done
done
done
