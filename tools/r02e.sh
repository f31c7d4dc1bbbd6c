set -eu
O=gpurun_out/r02e; mkdir -p $O
# one small case first: a fault must not be repeated
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "test_scan_example_map or random_poses" > $O/t0.log 2>&1 || { tail -20 $O/t0.log; exit 1; }
tail -2 $O/t0.log
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q > $O/t1.log 2>&1 || { tail -30 $O/t1.log; exit 1; }
tail -2 $O/t1.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t2.log 2>&1 || { tail -30 $O/t2.log; exit 1; }
tail -2 $O/t2.log
