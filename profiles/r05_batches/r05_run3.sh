cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/ubench/swizzle_probe > gpurun_out/r05_swizzle_probe.txt 2>&1 || { tail -5 gpurun_out/r05_swizzle_probe.txt; exit 1; }
grep rate gpurun_out/r05_swizzle_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_swz_pmc -o p -- $GRAFT_REPO_ROOT/tools/ubench/swizzle_probe > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
fs=glob.glob('gpurun_out/r05_swz_pmc/**/*counter_collection.csv',recursive=True)
rows=[]
for f in fs:
    rows+=list(csv.DictReader(open(f)))
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(list))
for r in rows:
    if 'rate' in r['Kernel_Name']:
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out=open('gpurun_out/r05_swz_pmc.txt','w')
for k in agg:
    line=k[:60]+' '+' '.join('%s=%.0f'%(c,sum(v)/len(v)) for c,v in agg[k].items())
    print(line); out.write(line+'\n')
PY
