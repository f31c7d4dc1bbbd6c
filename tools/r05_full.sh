cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q > gpurun_out/r05_fullsize.txt 2>&1; rc=$?
tail -8 gpurun_out/r05_fullsize.txt
exit $rc
