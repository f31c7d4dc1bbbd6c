#!/bin/bash
set -u
O=gpurun_out/r03g; mkdir -p $O
T="timeout -k 10 500"
$T python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "identical or batched or autoreset or odd_configs or closed_loop_golden or hipgraph" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
$T python tools/group_sweep.py 1024 2048 4096 8192 16384 32768 65536 --paths classic,closed > $O/closed_sweep.txt 2>&1
$T python tools/graph_vs_eager.py 4096 65536 > $O/graph_vs_eager.txt 2>&1
tail -n +1 $O/*.txt | grep -v amdgpu.ids
