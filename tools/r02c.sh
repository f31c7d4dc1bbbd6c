set -u
O=gpurun_out/r02c; mkdir -p $O
python -m pytest tests/test_gpu_step.py -x -q -k "hipgraph" > $O/graphtests.log 2>&1; echo "graph tests rc=$?"; tail -3 $O/graphtests.log
./tools/ubench/gather_cost2 > $O/gather_cost2.txt 2>&1; cat $O/gather_cost2.txt
python tools/graph_vs_eager.py > $O/graph_vs_eager.txt 2>&1; cat $O/graph_vs_eager.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_eager -o t -- python3 $R/tools/graph_vs_eager.py eager 65536 > $R/$O/trace_eager.out 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_graph -o t -- python3 $R/tools/graph_vs_eager.py graph_nocopy 65536 > $R/$O/trace_graph.out 2>&1
cd $R
for m in eager graph; do f=$(find $O/trace_$m -name "*kernel_trace.csv" | head -1); echo "== $m $f"; python tools/trace_gaps.py $f 400 > $O/gaps_$m.txt 2>&1; cat $O/gaps_$m.txt; done
rm -rf $O/trace_eager $O/trace_graph
python tools/sweep.py > $O/sweep.txt 2>&1
F110_LIB=$R/build_variants/nopark.so python tools/sweep.py >> $O/sweep.txt 2>&1
python tools/sweep.py >> $O/sweep.txt 2>&1
F110_LIB=$R/build_variants/nopark.so python tools/sweep.py >> $O/sweep.txt 2>&1
for w in 1 2 4 8; do echo "WPC=$w" >> $O/sweep.txt; F110_WPC=$w python tools/sweep.py --envs 4096 --steps 200 >> $O/sweep.txt 2>&1; done
for s in "*:0" "*:1" "2048:0,*:1" "1024:0,*:2" "*:0,3072:2" "*:0,1024:1,2048:2" "*:1,2048:2"; do echo "STAGES=$s" >> $O/sweep.txt; F110_STAGES="$s" python tools/sweep.py --envs 4096 --steps 200 >> $O/sweep.txt 2>&1; done
grep -v amdgpu.ids $O/sweep.txt
