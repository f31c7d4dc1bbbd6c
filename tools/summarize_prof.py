"""Condenses a tools/profile.sh output directory into the text committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
for f in find('stats/**/*kernel_stats.csv') + find('bitmap_stats/**/*kernel_stats.csv'):
    for row in csv.DictReader(open(f)):
        print('%-70s calls %6s  avg_us %10.2f  total_ms %10.3f  pct %6s' % (
            row['Name'][:70], row['Calls'], float(row['AverageNs']) / 1e3, float(row['TotalDurationNs']) / 1e6,
            row['Percentage']))
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
        if ('bitmap_kernel' if 'bmwrite' in f else 'scan_kernel') not in row['Kernel_Name']:
            continue
        acc[row['Counter_Name']] += float(row['Counter_Value'])
        cnt[row['Counter_Name']] += 1
    for k in sorted(acc):
        print('%-14s %-34s %18.1f  (n=%d)' % (os.path.basename(os.path.dirname(os.path.dirname(f)))[:14], k, acc[k] / cnt[k], cnt[k]))
