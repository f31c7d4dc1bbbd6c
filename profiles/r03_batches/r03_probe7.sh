#!/bin/bash
set -u
O=gpurun_out/r03k; mkdir -p $O
T="timeout -k 10 600"
$T python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "two_agents or raycast or batched or fuzz or odd_configs or fullsize" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for cfg in "65536 1" "4096 1" "16384 2"; do
  set -- $cfg
  for ev in 1 4 0; do
    if [ $ev = 0 ]; then extra="--no-scan-events"; else extra="--scan-events-every $ev"; fi
    $T python bench.py --envs $1 --agents $2 --steps 20 --warmup 5 --no-cpu-baseline $extra > $O/b.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('$O/b.json')); r=d.get('roofline') or {}
print('$1x$2 events-every $ev driver-form: %.2f M/s %.4f ms/step  scan %s ms (n=%s)  sustained %.2f M/s' % (d['value']/1e6, d['ms_per_step'], r.get('avg_launch_ms'), r.get('launches'), d['sustained']['value']/1e6))" >> $O/event_cost.txt
  done
done
cat $O/event_cost.txt
