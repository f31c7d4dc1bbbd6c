set -eu
O=gpurun_out/r02k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -1 $O/t.log
python bench.py --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; cut -c1-330 $O/bench_driverform.json
python bench.py --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline > $O/bench_driverform_nospinup.json 2>/dev/null; cut -c1-330 $O/bench_driverform_nospinup.json
python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null; cut -c1-330 $O/bench_default.json
python bench.py --no-cpu-baseline --envs 4096 > $O/bench_4096x1.json 2>/dev/null; cut -c1-330 $O/bench_4096x1.json
python bench.py --no-cpu-baseline --envs 16384 --agents 2 > $O/bench_16384x2.json 2>/dev/null; cut -c1-330 $O/bench_16384x2.json
F110_BENCH_ONE_DEVICE=1 F110_BENCH_BACKEND=gloo python bench.py --gpus 2 --envs 32768 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/rehearse2.json; cut -c1-330 $O/rehearse2.json
