cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_reference_kats.py tests/test_gpu_mirrors.py -x -q > gpurun_out/r05_quick_test.txt 2>&1; rc=$?
tail -3 gpurun_out/r05_quick_test.txt
[ $rc -eq 0 ] || exit $rc
for m in maps/berlin maps/skirk maps/vegas; do for r in 1 2; do F110_LIB_OLDER=1 F110_LIB=variants_ship/base.so timeout -k 10 120 python tools/sweep_map.py $m 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/round4 /"; timeout -k 10 120 python tools/sweep_map.py $m 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/round5 /"; done; done | tee gpurun_out/r05_other_maps.txt
