#!/bin/bash
# stage lists (waves per car along the launch) at 4 096 cars: F110_STAGES sweeps, tools/sweep.py
mkdir -p gpurun_out/st4096
for rep in 1 2; do
for st in "default" "*:1" "*:2" "*:0,3072:2" "*:1,2048:2" "*:1,1024:3" "2048:1,*:2" "*:0,2048:3" "*:1,2048:3" "1024:0,*:1,1024:2"; do
  if [ "$st" = default ]; then unset F110_STAGES; else export F110_STAGES="$st"; fi
  echo -n "stages $st: " | tee -a gpurun_out/st4096/log.txt
  timeout -k 10 120 python tools/sweep.py --envs 4096 --steps 300 --warmup 100 2>&1 | tail -1 | tee -a gpurun_out/st4096/log.txt
done
done
