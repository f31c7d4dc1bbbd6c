#!/bin/bash
set -e
mkdir -p gpurun_out/bm2
timeout -k 10 600 python -m pytest tests/test_gpu_bitmap.py -x -q > gpurun_out/bm2/tests.log 2>&1 || { tail -30 gpurun_out/bm2/tests.log; exit 1; }
tail -3 gpurun_out/bm2/tests.log
for m in FILL POLYGON RAYS; do timeout -k 10 200 python tools/bench_bitmap.py --mode $m 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bm2/bench.log; done
timeout -k 10 200 python tools/bench_bitmap.py --mode FILL --channels 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bm2/bench.log
