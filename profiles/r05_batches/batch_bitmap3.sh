#!/bin/bash
set -e
mkdir -p gpurun_out/bm5
for rep in 1 2; do
for v in tree oldbm d4s1 d8s1 d6s2 d12s1 d8s2; do
  echo "## $v" | tee -a gpurun_out/bm5/bench.log
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  timeout -k 10 200 python tools/bench_bitmap.py --mode FILL 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bm5/bench.log
done
done
