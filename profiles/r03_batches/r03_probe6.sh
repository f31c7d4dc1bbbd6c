#!/bin/bash
set -u
O=gpurun_out/r03i; mkdir -p $O
T="timeout -k 10 600"
$T python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "two_agents or raycast or batched or fuzz or odd_configs or fullsize" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
$T python bench.py --envs 16384 --agents 2 --no-cpu-baseline > $O/bench_16384x2.json 2> $O/bench_16384x2.err; cat $O/bench_16384x2.json | cut -c1-400
for s in 0 1; do $T python tools/region_order_probe.py --sort $s >> $O/region_probe.txt 2>&1; done
$T python tools/region_order_probe.py --sort 1 --region 32 >> $O/region_probe.txt 2>&1
$T python tools/region_order_probe.py --sort 1 --region 128 >> $O/region_probe.txt 2>&1
$T python tools/region_order_probe.py --sort 1 --every 1 >> $O/region_probe.txt 2>&1
export TMPDIR=/tmp; cd /tmp
for s in 0 1; do
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sort$s -o pmc -- python3 $GRAFT_REPO_ROOT/tools/region_order_probe.py --sort $s --steps 16 > $GRAFT_REPO_ROOT/$O/pmc_sort$s.out 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for s in (0, 1):
    for f in glob.glob('gpurun_out/r03i/pmc_sort%d/**/*counter_collection.csv' % s, recursive=True):
        acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            if 'scan_kernel' in row['Kernel_Name']:
                acc[row['Counter_Name']] += float(row['Counter_Value']); cnt[row['Counter_Name']] += 1
        print('sort', s, {k: round(acc[k] / cnt[k]) for k in acc}, 'n', dict(cnt))
PY
grep -v amdgpu.ids $O/region_probe.txt
