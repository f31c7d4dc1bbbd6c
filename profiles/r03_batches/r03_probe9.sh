#!/bin/bash
set -u
O=gpurun_out/r03r; mkdir -p $O
T="timeout -k 10 300"
run() { # envs agents stages lr
  echo -n "$1x$2 stages=$3 long_rays_first=$4 : " >> $O/long_rays2.txt
  if [ "$3" = default ]; then F110_LONG_RAYS_FIRST=$4 $T python tools/sweep.py --envs $1 --agents $2 --steps 120 --warmup 60 2>&1 | grep -v amdgpu.ids >> $O/long_rays2.txt
  else F110_STAGES="$3" F110_LONG_RAYS_FIRST=$4 $T python tools/sweep.py --envs $1 --agents $2 --steps 120 --warmup 60 2>&1 | grep -v amdgpu.ids >> $O/long_rays2.txt; fi
}
for st in default '2048:2,*:0' '1024:2,*:0,1024:2' '*:2' '512:2,*:0,1536:2'; do for lr in 0 8,120 8,80 8,160; do run 4096 1 "$st" $lr; done; done
for st in default '2048:2,*:0,2048:2' '4096:2,*:0,2048:2' '1024:2,*:0,2048:2'; do for lr in 0 8,120 8,160; do run 65536 1 "$st" $lr; done; done
for st in default '2048:2,*:0,2048:2' '4096:2,*:0,2048:2'; do for lr in 0 8,120 8,80; do run 16384 2 "$st" $lr; done; done
cat $O/long_rays2.txt
