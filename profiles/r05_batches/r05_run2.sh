cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_a.txt 2>&1; rc=$?
tail -15 gpurun_out/r05_gputest_a.txt
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
F110_LIB=variants_ship/base.so timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 >> gpurun_out/r05_sweep_a.txt 2>&1 || exit 1
timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 >> gpurun_out/r05_sweep_a.txt 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/r05_sweep_a.txt
