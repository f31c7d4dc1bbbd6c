#!/bin/bash
set -e
mkdir -p gpurun_out/bm25
timeout -k 10 600 python -m pytest tests/test_gpu_bitmap.py -x -q > gpurun_out/bm25/tests.log 2>&1 || { tail -30 gpurun_out/bm25/tests.log; exit 1; }
tail -1 gpurun_out/bm25/tests.log
for rep in 1 2 3; do
for v in tree prev; do
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  echo -n "$v: " | tee -a gpurun_out/bm25/bench.log
  timeout -k 10 200 python tools/bench_bitmap.py --mode FILL --reps 100 2>&1 | grep "^bitmap" | tee -a gpurun_out/bm25/bench.log
done
done
F110_LIB=$PWD/variants_ship/bmtl.so F110_LIB_OLDER=1 timeout -k 10 200 python tools/bench_bitmap.py --mode FILL --reps 2 2>&1 | grep "timeline" | tail -1 | tee -a gpurun_out/bm25/bench.log
