#!/bin/bash
# builds kernel variants for tools/sweep.py: build_variants/<name>.so
#   WAVES="2 4" REFILL="36 40 44" tools/build_variants.sh
set -e
cd "$(dirname "$0")/.."
for w in ${WAVES:-2 4}; do for r in ${REFILL:-36 40 44}; do
  tools/build_variant.sh w${w}_r${r} -DF110_SCAN_WAVES=$w -DF110_REFILL_MIN_IDLE=$r
done; done
ls build_variants
