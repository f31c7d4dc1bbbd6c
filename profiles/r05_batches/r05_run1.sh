cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in base noclamp extra2 extra4; do F110_LIB=variants_ship/$v.so timeout -k 10 120 python tools/sweep.py --steps 150 --warmup 100 >> gpurun_out/r05_valu_sens.txt 2>&1 || exit 1; done; done
cat gpurun_out/r05_valu_sens.txt
