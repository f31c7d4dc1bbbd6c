#!/bin/bash
set -e
mkdir -p gpurun_out/bm9
for rep in 1 2 3; do
for v in tree prev oldbm; do
  echo "## $v" | tee -a gpurun_out/bm9/bench.log
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  timeout -k 10 200 python tools/bench_bitmap.py --mode FILL --reps 100 2>&1 | grep "^bitmap" | tee -a gpurun_out/bm9/bench.log
done
done
