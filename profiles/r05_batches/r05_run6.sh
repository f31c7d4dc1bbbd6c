cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_c.txt 2>&1; rc=$?
tail -3 gpurun_out/r05_gputest_c.txt
[ $rc -eq 0 ] || exit $rc
F110_LIB=$GRAFT_REPO_ROOT/variants_ship/bounds.so F110_CHECK_DEVICE_ERRORS=1 timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not rccl" > gpurun_out/r05_gputest_bounds.txt 2>&1; rc=$?
tail -3 gpurun_out/r05_gputest_bounds.txt
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh r05b > gpurun_out/prof_r05b.log 2>&1
tail -26 gpurun_out/prof_r05b.log
F110_LIB=variants_ship/timeline.so timeout -k 10 200 python tools/timeline.py --envs 65536 > gpurun_out/r05_timeline_65536.txt 2>&1
grep "per car" gpurun_out/r05_timeline_65536.txt
