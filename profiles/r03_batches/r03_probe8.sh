#!/bin/bash
set -u
O=gpurun_out/r03q; mkdir -p $O
T="timeout -k 10 900"
$T python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for cfg in "4096 1" "65536 1" "16384 2" "8192 1"; do
  set -- $cfg
  for lr in 0 "8,120" "8,80" "8,160" "4,120" "16,120" "8,60"; do
    echo -n "$1x$2 long_rays_first=$lr : " >> $O/long_rays.txt
    F110_LONG_RAYS_FIRST=$lr $T python tools/sweep.py --envs $1 --agents $2 --steps 120 --warmup 60 2>&1 | grep -v amdgpu.ids >> $O/long_rays.txt
  done
done
cat $O/long_rays.txt
