cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_bitmap.txt; rm -f $out
for r in 1 2; do for v in default bm256 bm384; do
  echo "## $v" >> $out
  if [ $v = default ]; then timeout -k 10 120 python tools/bench_bitmap.py >> $out 2>&1 || exit 1
  else F110_LIB=variants_ship/$v.so timeout -k 10 120 python tools/bench_bitmap.py >> $out 2>&1 || exit 1; fi
done; done
grep -v "amdgpu.ids\|occupancy" $out
