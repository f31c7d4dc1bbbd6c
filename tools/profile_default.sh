#!/bin/bash
# rocprofv3 kernel trace of the 200-step default bench run (run on the GPU box through gpurun): scan_kernel over the 200 timed dispatches
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
O=$ROOT/gpurun_out/prof_r05_default
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --sustained 0 --steady-state 0 > $O.json 2> $O.err
python3 - <<PY
import csv,glob,json
f=glob.glob('$O/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'scan_kernel' in r['Kernel_Name'] and 'post_scan' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
last=d[-200:]
print('scan_kernel dispatches %d; last 200 (the timed region of the default run): mean %.2f us, min %.2f, max %.2f' % (len(d), sum(last)/len(last), min(last), max(last)))
j=json.load(open('$O.json')); r=j['roofline']
print('bench line under rocprof: value %.2f M, ms/step %.4f, scan by events %.4f ms (n=%d), frac %.3f' % (j['value']/1e6, j['ms_per_step'], r['avg_launch_ms'], r['launches'], r['frac']))
PY
