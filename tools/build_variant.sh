#!/bin/bash
# one named build variant of the library: tools/build_variant.sh <name> [-DMACRO=VALUE ...]  -> build_variants/<name>.so
# (build_variants/ is git-ignored and, via .gpurunignore, not shipped; ship a variant by building it under gpurun_out/)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=${VARIANT_DIR:-build_variants}
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value "$@" -o $out/$name.so red_gym_amd/csrc/f110_abi.hip
echo $out/$name.so
