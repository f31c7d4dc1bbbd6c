# A/B sweep of library variants on one box: tools/r05_sweepv.sh <out-name> <variant> ... ("default" = the tree's library)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
rm -f $out
for r in 1 2; do for v in "$@"; do
  if [ $v = default ]; then timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 $SWEEP_ARGS >> $out 2>&1 || exit 1
  else F110_LIB=variants_ship/$v.so timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 $SWEEP_ARGS >> $out 2>&1 || exit 1; fi
done; done
grep -v amdgpu.ids $out
