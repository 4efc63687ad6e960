#!/bin/bash
# Development helper: build an A/B variant of the library with extra -D flags for ONE translation unit.
#   scripts/build_variant.sh <tag> <file.hip> [-DFOO=1 ...]   ->  cyten_amd/lib/libcyten_amd_<tag>.so
# Run both variants inside the same gpurun call with CYTEN_AMD_LIB=... (devices differ between boxes).
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -m cyten_amd.build >/dev/null
mkdir -p /tmp/cyb_variant_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Iinclude -Icyten_amd/csrc "$@" \
    -c cyten_amd/csrc/$src -o /tmp/cyb_variant_$tag/${src%.hip}.o
objs=""
for o in cyten_amd/lib/obj/*.o; do
  b=$(basename $o)
  if [ "$b" == "${src%.hip}.o" ]; then objs="$objs /tmp/cyb_variant_$tag/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o cyten_amd/lib/libcyten_amd_$tag.so $objs
echo cyten_amd/lib/libcyten_amd_$tag.so
