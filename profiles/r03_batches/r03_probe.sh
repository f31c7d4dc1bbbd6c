#!/bin/bash
# round-3 probe batch (one gpurun call): where does a 4 096-car step spend its time?
set -u
O=gpurun_out/r03b; mkdir -p $O
T="timeout -k 10 240"
export F110_LIB=$PWD/variants_ship/timeline.so
$T python tools/timeline.py --envs 4096 --path classic > $O/tl_classic_default.txt 2>&1
$T python tools/timeline.py --envs 4096 --path classic --stages '*:0' > $O/tl_classic_whole.txt 2>&1
$T python tools/timeline.py --envs 4096 --path classic --stages '*:2' > $O/tl_classic_quarter.txt 2>&1
F110_GROUP_NOFUSE=1 $T python tools/timeline.py --envs 4096 --path group:4 > $O/tl_group4_nofuse.txt 2>&1
$T python tools/timeline.py --envs 4096 --path group:4 > $O/tl_group4_fused.txt 2>&1
$T python tools/timeline.py --envs 65536 --path classic > $O/tl_classic_65536.txt 2>&1
unset F110_LIB
$T python tools/stage_sweep.py --envs 4096 default '*:0' '*:1' '*:2' '3072:0,*:2' '1024:0,*:2' '2048:0,*:1' '1024:0,*:1' '*:3' '2048:1,*:2' > $O/stages_4096.txt 2>&1
$T python tools/stage_sweep.py --envs 8192 default '*:0' '*:1' '*:2' '4096:0,*:1' '6144:0,*:2' > $O/stages_8192.txt 2>&1
for pad in 0 5000 8000 12000 18000; do
  F110_SCAN_PAD_LDS=$pad F110_GROUP=0 $T python tools/sweep.py --envs 4096 --steps 200 >> $O/pad_4096.txt 2>&1
  F110_SCAN_PAD_LDS=$pad F110_GROUP=0 $T python tools/sweep.py --envs 65536 --steps 60 >> $O/pad_65536.txt 2>&1
done
F110_GROUP_NOFUSE=1 $T python tools/group_sweep.py 2048 4096 8192 --paths classic,group:2,group:4,group:8 > $O/group_nofuse.txt 2>&1
for v in gmin6 gmin5 gmin4; do
  echo $v >> $O/group_fused_variants.txt
  F110_LIB=$PWD/variants_ship/$v.so $T python tools/group_sweep.py 4096 --paths classic,group:4 >> $O/group_fused_variants.txt 2>&1
done
tail -n +1 $O/*.txt | grep -v amdgpu.ids
