cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_sweep_c.txt; rm -f $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 >> $out 2>&1 || exit 1; }
for r in 1 2; do
run X=default
for v in r44 r48 r52 r56; do run F110_LIB=variants_ship/$v.so; done
for s in '*:0,1024:2' '*:0,3072:2' '*:0,2048:1' '*:0,4096:1' '*:0,1024:3' '*:0,1024:2,1024:3'; do run F110_STAGES="$s"; done
for s in '*:0,1024:2' '*:0,2048:2' '*:0,3072:2' ; do run F110_STAGES="$s" F110_LIB=variants_ship/r48.so; done
done
grep -v amdgpu.ids $out
