cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_noise.py -x -q > gpurun_out/r05_noise_test.txt 2>&1; rc=$?
tail -25 gpurun_out/r05_noise_test.txt
exit $rc
