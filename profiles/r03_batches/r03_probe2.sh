#!/bin/bash
# round-3 probe batch 2: timelines of a 4 096-car launch, drain priority, graph forms, new planner tests
set -u
O=gpurun_out/r03c; mkdir -p $O gpurun_out/graphs
T="timeout -k 10 300"
$T python -m pytest tests/test_gpu_mirrors.py -x -q -m gpu -k "pure_pursuit" > $O/planner_tests.log 2>&1; echo "planner tests rc=$?"
export F110_LIB=$PWD/variants_ship/timeline.so
$T python tools/timeline.py --envs 4096 --path classic > $O/tl_classic_default.txt 2>&1
$T python tools/timeline.py --envs 4096 --path classic --stages '*:0' > $O/tl_classic_whole.txt 2>&1
$T python tools/timeline.py --envs 4096 --path classic --stages '*:2' > $O/tl_classic_quarter.txt 2>&1
F110_GROUP_NOFUSE=1 $T python tools/timeline.py --envs 4096 --path group:4 > $O/tl_group4_nofuse.txt 2>&1
$T python tools/timeline.py --envs 65536 --path classic > $O/tl_classic_65536.txt 2>&1
unset F110_LIB
for v in prio1 prio3; do
  for n in 4096 65536; do
    echo "$v $n" >> $O/prio.txt
    F110_LIB=$PWD/variants_ship/$v.so $T python tools/sweep.py --envs $n --steps 100 >> $O/prio.txt 2>&1
  done
done
for n in 4096 65536; do $T python tools/sweep.py --envs $n --steps 100 >> $O/prio.txt 2>&1; done
$T python tools/graph_vs_eager.py > $O/graph_vs_eager.txt 2>&1
tail -n +1 $O/*.txt $O/planner_tests.log | grep -v amdgpu.ids
ls -la gpurun_out/graphs
