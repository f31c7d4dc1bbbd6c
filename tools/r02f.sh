set -eu
O=gpurun_out/r02f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q -k "stage_lists or multi_car" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -1 $O/t.log
: > $O/sweep.txt
for s in "*:0,2048:2" "*:-1,4096:0,2048:2" "*:-2,6144:0,2048:2" "*:-2,12288:0,2048:2" "*:-3,8192:0,2048:2" "*:-2,8192:-1,4096:0,2048:2" "*:-3,8192:-1,4096:0,2048:2"; do echo "STAGES=$s" >> $O/sweep.txt; F110_STAGES="$s" python tools/sweep.py >> $O/sweep.txt 2>&1; done
grep -v amdgpu.ids $O/sweep.txt
