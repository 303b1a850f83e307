#!/bin/bash
# Builds dbg/libdudf_<tag>.so with extra -D flags for ONE translation unit (timing experiments with debug knobs):
#   bash tools/build_dbg.sh <tag> <unit: sweep_bf16|wgrad|...> "-DDUDF_SWEEP_DBG=3"
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; unit=$2; flags=$3
mkdir -p "$R/dbg"
B=$R/diffudf_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$R/diffudf_amd/csrc/dudf_$unit.hip" -o "$R/dbg/${unit}_$tag.o" || exit 1
objs=""
for u in sweep sweep_bf16 wgrad misc sample capudf api; do
  if [ "$u" = "$unit" ]; then objs="$objs $R/dbg/${unit}_$tag.o"; else objs="$objs $B/dudf_$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/dbg/libdudf_$tag.so" $objs
