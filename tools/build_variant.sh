#!/bin/bash
# one named build variant of the library: tools/build_variant.sh <name> [-DMACRO=VALUE ...]  -> build_variants/<name>.so
# (build_variants/ is git-ignored and, via .gpurunignore, not shipped; ship a variant with VARIANT_DIR=variants_ship)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=${VARIANT_DIR:-build_variants}
mkdir -p $out
python3 - "$out" "$name" "$@" <<'PY'
import sys
sys.path.insert(0, '.')
from red_gym_amd import build
out, name, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
print(build.build(force=True, extra_flags=flags, lib='%s/%s.so' % (out, name), obj_dir='%s/.obj_%s' % (out, name)))
PY
