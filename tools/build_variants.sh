#!/bin/bash
# builds kernel variants for tools/sweep.py: build_variants/<name>.so
#   WAVES="2 4" REFILL="36 40 44" tools/build_variants.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value"
build() { name=$1; shift; hipcc $FLAGS "$@" -o build_variants/$name.so red_gym_amd/csrc/f110_abi.hip & }
n=0
for w in ${WAVES:-2 4}; do for r in ${REFILL:-36 40 44}; do
  build w${w}_r${r} -DF110_SCAN_WAVES=$w -DF110_REFILL_MIN_IDLE=$r; n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
done; done
wait
ls build_variants
