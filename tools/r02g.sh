set -eu
O=$PWD/gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp
pmc() { name=$1; shift; F110_STAGES="$SPEC" rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o pmc -- python3 $R/tools/sweep.py --steps 6 --warmup 30 > $O/pmc_$name.out 2> $O/pmc_$name.err; }
SPEC="*:0,2048:2"
pmc a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM
SPEC="*:-2,6144:0,2048:2"
pmc b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for tag in ('a', 'b'):
    f = glob.glob(os.path.join('gpurun_out/r02g/pmc_%s' % tag, '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'scan_kernel' in k:
            print(tag, k, {c: '%.1fM' % (sum(v[-6:]) / len(v[-6:]) / 1e6) for c, v in d.items()})
PY
rm -rf gpurun_out/r02g/pmc_a gpurun_out/r02g/pmc_b
