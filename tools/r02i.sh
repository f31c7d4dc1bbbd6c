set -eu
O=gpurun_out/r02i; mkdir -p $O; : > $O/bisect.txt
for k in d s e ds se dse; do echo "ONLY=$k" >> $O/bisect.txt; F110_LIB=$PWD/build_variants/dbg.so F110_DEBUG_ONLY=$k python tools/graph_vs_eager.py eager graph_nocopy 65536 >> $O/bisect.txt 2>&1; done
grep -v amdgpu.ids $O/bisect.txt
