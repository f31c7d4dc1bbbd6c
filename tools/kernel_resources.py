"""Compiles the library with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
    python tools/kernel_resources.py [filter] [-- extra hipcc flags]
(VGPRs, SGPRs, spills, scratch, occupancy in waves per SIMD, LDS bytes).  Works without a GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--')
    args, extra = args[:i], args[i + 1:]
flt = args[0] if args else ''
sys.path.insert(0, ROOT)
from red_gym_amd import build as _b  # noqa: E402
err = ''
for u in _b.UNITS:   # every translation unit of the library (device code only: -c of one unit at a time)
    cmd = ['/opt/rocm/bin/hipcc'] + _b.FLAGS + ['-Rpass-analysis=kernel-resource-usage', '--cuda-device-only', '-c', '-o', '/dev/null',
                                                os.path.join(_b.CSRC, u + '.hip')] + extra
    err += subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r'remark: (?:\S+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                  r'SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(' ')[0] + (' Spill' if 'Spill' in k else '')] = v
for r in rows:
    if flt and flt not in r['name']:
        continue
    print('%-70s VGPR %3s AGPR %2s SGPR %3s spill s%s/v%s scratch %4s occ %s LDS %s' % (
        r['name'][:70], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('SGPRs Spill'), r.get('VGPRs Spill'),
        r.get('ScratchSize'), r.get('Occupancy'), r.get('LDS')))
