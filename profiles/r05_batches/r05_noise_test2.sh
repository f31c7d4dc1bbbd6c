cd $GRAFT_REPO_ROOT
F110_LIB=$GRAFT_REPO_ROOT/variants_ship/prev.so timeout -k 10 300 python -m pytest tests/test_gpu_noise.py -x -q -k "prefetch_waits" > gpurun_out/r05_noise_test_prevlib.txt 2>&1
tail -6 gpurun_out/r05_noise_test_prevlib.txt
timeout -k 10 300 python -m pytest tests/test_gpu_noise.py -x -q > gpurun_out/r05_noise_test.txt 2>&1
tail -4 gpurun_out/r05_noise_test.txt
exit 0
