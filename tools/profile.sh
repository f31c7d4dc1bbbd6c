#!/bin/bash
# rocprofv3 passes for the step path (run on the GPU box through gpurun):
#   [ENVS=65536] [AGENTS=1] [BITMAP=1] tools/profile.sh <tag>
# 1. --kernel-trace --stats of the driver's bench command   -> gpurun_out/prof_<tag>/stats
# 2. separate --pmc passes (never combined with tracing domains) on tools/sweep.py, same config
# 3. (BITMAP=1) the scans' consumer: bitmap_kernel
set -u
TAG=${1:-r03}
ENVS=${ENVS:-65536}; AGENTS=${AGENTS:-1}; BITMAP=${BITMAP:-0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 20 --warmup 5 --sustained 0 --steady-state 0 --no-cpu-baseline --envs $ENVS --agents $AGENTS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats rc=$?"
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/tools/sweep.py --steps 6 --warmup 30 --envs $ENVS --agents $AGENTS > $OUT/pmc_$name.out 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pmc sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
if [ "$BITMAP" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bitmap_stats -o bm -- python3 $ROOT/tools/bench_bitmap.py --reps 10 > $OUT/bitmap_under_rocprof.txt 2> $OUT/bitmap_stats.err
  echo "bitmap stats rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_bmwrite -o pmc -- python3 $ROOT/tools/bench_bitmap.py --reps 3 > $OUT/pmc_bmwrite.out 2> $OUT/pmc_bmwrite.err
  echo "pmc bmwrite rc=$?"
fi
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
