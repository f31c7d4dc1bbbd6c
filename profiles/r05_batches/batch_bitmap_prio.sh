#!/bin/bash
set -e
mkdir -p gpurun_out/bmprio
for rep in 1 2 3; do
for v in tree p2 p3 p4; do
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  echo -n "$v: " | tee -a gpurun_out/bmprio/bench.log
  timeout -k 10 200 python tools/bench_bitmap.py --mode FILL --reps 100 2>&1 | grep "^bitmap" | tee -a gpurun_out/bmprio/bench.log
done
done
