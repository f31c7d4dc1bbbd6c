#!/bin/bash
set -u
O=gpurun_out/r03d; mkdir -p $O
T="timeout -k 10 400"
$T python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "identical or batched or autoreset or odd_configs or closed_loop_golden" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
$T python tools/group_sweep.py 1024 2048 4096 8192 16384 --paths classic,group:2,group:4,group:8 > $O/group_sweep.txt 2>&1
F110_LIB=$PWD/variants_ship/noprio.so $T python tools/group_sweep.py 4096 8192 --paths classic,group:2,group:4 > $O/group_sweep_noprio.txt 2>&1
F110_GROUP_FUSE_DYN=1 $T python tools/group_sweep.py 4096 --paths classic,group:2,group:4 > $O/group_sweep_fusedyn.txt 2>&1
export F110_LIB=$PWD/variants_ship/timeline.so
$T python tools/timeline.py --envs 4096 --path group:2 > $O/tl_group2.txt 2>&1
$T python tools/timeline.py --envs 4096 --path group:4 > $O/tl_group4.txt 2>&1
unset F110_LIB
tail -n +1 $O/*.txt | grep -v amdgpu.ids
