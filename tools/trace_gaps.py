"""Summarises a rocprofv3 --kernel-trace CSV: per kernel mean duration, and the mean idle gap between the end of
one dispatch and the start of the next (same queue order):  python tools/trace_gaps.py <..._kernel_trace.csv> [skip]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[skip:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    dur[a['Kernel_Name'][:60]].append(int(a['End_Timestamp']) - int(a['Start_Timestamp']))
    gap[a['Kernel_Name'][:28] + ' -> ' + b['Kernel_Name'][:28]].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
print('kernel durations (us):')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('  %-62s n=%5d mean %9.2f' % (k, len(v), sum(v) / len(v) / 1e3))
print('gaps end -> next start (us):')
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('  %-62s n=%5d mean %8.2f median %8.2f' % (k, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3))
tot = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('span %.3f ms, busy %.3f ms' % (tot / 1e6, sum(sum(v) for v in dur.values()) / 1e6))
