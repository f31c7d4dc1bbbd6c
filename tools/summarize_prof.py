"""Condenses a tools/profile.sh output directory into the text committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def is_scan(name):
    """f110::scan_kernel<...> only (post_scan_kernel, the env / opponent set-up launch of A > 1, also carries the words)"""
    return 'scan_kernel' in name and 'post_scan_kernel' not in name


out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for label, pat in (('bench.py', 'stats/**/*kernel_stats.csv'), ('tools/bench_bitmap.py', 'bitmap_stats/**/*kernel_stats.csv')):
    print('== kernel stats of %s (rocprofv3 --kernel-trace --stats) ==' % label)
    for f in find(pat):
        for row in csv.DictReader(open(f)):
            if float(row['Percentage']) < 0.05:
                continue
            print('%-70s calls %6s  avg_us %10.2f  total_ms %10.3f  pct %6s' % (
                row['Name'][:70], row['Calls'], float(row['AverageNs']) / 1e3, float(row['TotalDurationNs']) / 1e6,
                row['Percentage']))
    print()
# the bench's own events bracket the K TIMED launches only; the stats above average every launch of the process
# (spin-up, reset, warm-up too), so the same K dispatches are averaged from the trace for comparison
for f in find('stats/**/*kernel_trace.csv'):
    rows = [r for r in csv.DictReader(open(f)) if is_scan(r['Kernel_Name'])]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    K = 20
    if len(d) >= K:
        print('scan_kernel from the kernel trace: %d dispatches, mean %.2f us; the last %d (the timed region of '
              '`bench.py --steps %d`): mean %.2f us, min %.2f, max %.2f' % (len(d), sum(d) / len(d), K, K, sum(d[-K:]) / K, min(d[-K:]), max(d[-K:])))
print()
print('== bench line under rocprof ==')
for f in find('bench_under_rocprof.json'):
    print(open(f).read().strip()[-900:])
print()
for f in find('bitmap_under_rocprof.txt'):
    print(open(f).read().strip()[-600:])
print()
print('== PMC counters: mean per dispatch of scan_kernel (bmwrite: of bitmap_kernel) ==')
for f in find('pmc_*/**/*counter_collection.csv'):
    acc, cnt = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(f)):
        if not (('bitmap_kernel' in row['Kernel_Name']) if 'bmwrite' in f else is_scan(row['Kernel_Name'])):
            continue
        acc[row['Counter_Name']] += float(row['Counter_Value'])
        cnt[row['Counter_Name']] += 1
    for k in sorted(acc):
        print('%-14s %-34s %18.1f  (n=%d)' % (os.path.basename(os.path.dirname(os.path.dirname(f)))[:14], k, acc[k] / cnt[k], cnt[k]))
