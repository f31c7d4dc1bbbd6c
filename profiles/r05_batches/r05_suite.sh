cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_d.txt 2>&1; rc=$?
tail -4 gpurun_out/r05_gputest_d.txt
[ $rc -eq 0 ] || exit $rc
F110_LIB=$GRAFT_REPO_ROOT/variants_ship/bounds.so F110_CHECK_DEVICE_ERRORS=1 timeout -k 10 1000 python -m pytest tests -m gpu -q -x -k "not rccl" > gpurun_out/r05_gputest_bounds.txt 2>&1; rc=$?
tail -3 gpurun_out/r05_gputest_bounds.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
