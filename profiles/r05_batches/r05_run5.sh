cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_sweep_d.txt; rm -f $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 $ARGS >> $out 2>&1 || exit 1; }
for r in 1 2; do
ARGS="--envs 4096" run F110_LIB=variants_ship/base.so; ARGS="--envs 4096" run X=default
ARGS="--envs 16384 --agents 2" run F110_LIB=variants_ship/base.so; ARGS="--envs 16384 --agents 2" run X=default
ARGS="--envs 16384" run F110_LIB=variants_ship/base.so; ARGS="--envs 16384" run X=default
done
grep -v amdgpu.ids $out
