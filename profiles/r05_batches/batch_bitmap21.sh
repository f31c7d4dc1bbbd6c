#!/bin/bash
set -e
mkdir -p gpurun_out/bm21
timeout -k 10 900 python -m pytest tests/test_gpu_bitmap.py tests/test_gpu_step.py tests/test_gpu_mirrors.py -x -q > gpurun_out/bm21/tests.log 2>&1 || { tail -30 gpurun_out/bm21/tests.log; exit 1; }
tail -1 gpurun_out/bm21/tests.log
for rep in 1 2; do
for args in "--mode FILL --channels 4" "--mode FILL"; do
for v in tree oldbm; do
  if [ $v = tree ]; then unset F110_LIB F110_LIB_OLDER; else export F110_LIB=$PWD/variants_ship/$v.so F110_LIB_OLDER=1; fi
  echo -n "$v: " | tee -a gpurun_out/bm21/bench.log
  timeout -k 10 200 python tools/bench_bitmap.py $args --reps 100 2>&1 | grep "^bitmap" | tee -a gpurun_out/bm21/bench.log
done
done
done
unset F110_LIB F110_LIB_OLDER
timeout -k 10 300 python bench.py --bitmap FILL > gpurun_out/bm21/bench_bitmap.json 2> gpurun_out/bm21/bench_bitmap.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bm21/bench_bitmap.json').read().strip().splitlines()[-1])
print('bench --bitmap FILL:', round(d['value'] / 1e6, 2), 'M env-steps/s', d['ms_per_step'])
PY
