set -u
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q -k "stage_lists or multi_car or hipgraph" > $O/tests.log 2>&1; echo "stage tests rc=$?"; tail -5 $O/tests.log
python tools/graph_vs_eager.py > $O/graph_vs_eager.txt 2>&1; grep -v amdgpu.ids $O/graph_vs_eager.txt
R=$PWD
F110_LIB=$R/build_variants/base.so python tools/sweep.py > $O/sweep.txt 2>&1
for s in "" "*:0,2048:2" "*:-1,4096:0,2048:2" "*:-2,6144:0,2048:2" "*:-2,2048:2" "*:-2,12288:0,2048:2" "*:-3,8192:0,2048:2" "*:-2,8192:-1,4096:0,2048:2"; do echo "STAGES=$s" >> $O/sweep.txt; F110_STAGES="$s" python tools/sweep.py >> $O/sweep.txt 2>&1; done
F110_LIB=$R/build_variants/base.so python tools/sweep.py >> $O/sweep.txt 2>&1
grep -v amdgpu.ids $O/sweep.txt
