#!/bin/bash
# builds kernel variants for tools/sweep.py: build_variants/<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value"
build() { name=$1; shift; hipcc $FLAGS "$@" -o build_variants/$name.so red_gym_amd/csrc/f110_abi.hip & }
for w in ${WAVES:-4 8}; do for r in ${REFILL:-16 24 32}; do build w${w}_r${r} -DF110_SCAN_WAVES=$w -DF110_REFILL_MIN_IDLE=$r; done; wait; done
ls build_variants
