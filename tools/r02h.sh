set -eu
O=gpurun_out/r02h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -1 $O/t.log
python tools/graph_overhead_probe.py > $O/graph_probe.txt 2>&1; grep -v amdgpu.ids $O/graph_probe.txt
python tools/graph_vs_eager.py > $O/graph_vs_eager.txt 2>&1; grep -v amdgpu.ids $O/graph_vs_eager.txt
python tools/sweep.py > $O/sweep.txt 2>&1; grep -v amdgpu.ids $O/sweep.txt
