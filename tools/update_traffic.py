"""Refreshes profiles/traffic.json (what bench.py reports as roofline.traffic) from the --pmc FETCH_SIZE / WRITE_SIZE
passes of tools/profile.sh:  python tools/update_traffic.py gpurun_out/prof_<tag> <envs>x<agents> <summary name in profiles/>
HBM bytes per scan_kernel launch = 2 * FETCH_SIZE KB (gfx950 under-reports wide reads by up to 2x: upper bound) + WRITE_SIZE KB."""
import csv
import glob
import json
import os
import sys


def is_scan(name):
    """f110::scan_kernel<...> only (post_scan_kernel, the env / opponent set-up launch of A > 1, also carries the words)"""
    return 'scan_kernel' in name and 'post_scan_kernel' not in name


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof, key, summary = sys.argv[1], sys.argv[2], sys.argv[3]
val = {}
for name in ('fetch', 'write'):
    acc = cnt = 0
    for f in glob.glob(os.path.join(prof, 'pmc_%s' % name, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if is_scan(row['Kernel_Name']):
                acc += float(row['Counter_Value']); cnt += 1
    val[name] = acc / max(cnt, 1)
tj_path = os.path.join(ROOT, 'profiles', 'traffic.json')
tj = json.load(open(tj_path))
envs, agents = key.split('x')
tj[key] = {'hbm_bytes_per_launch': int(round((2 * val['fetch'] + val['write']) * 1024)), 'fetch_size_kb': round(val['fetch'], 1),
           'write_size_kb': round(val['write'], 1),
           'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/sweep.py (%s envs x %s agent(s)), mean per '
                   'scan_kernel dispatch (profiles/%s). FETCH_SIZE doubled per the gfx950 under-report rule (upper bound: 2-byte '
                   'gathers); writes are the fp32 scan tensor + state.' % (envs, agents, summary)}
json.dump(tj, open(tj_path, 'w'), indent=1)
print(key, tj[key])
