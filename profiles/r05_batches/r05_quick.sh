# quick check of a kernel change: parity + step tests, then an A/B sweep against variants_ship/prev.so
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_reference_kats.py -x -q > gpurun_out/r05_quick_test.txt 2>&1; rc=$?
tail -3 gpurun_out/r05_quick_test.txt
[ $rc -eq 0 ] || exit $rc
bash tools/r05_sweepv.sh r05_quick_sweep.txt prev default
