#!/bin/bash
# Builds dbg/libdudf_<tag>.so with extra -D flags for EVERY translation unit:  bash tools/build_all_dbg.sh <tag> "-DDUDF_P24_ARRAYS=2"
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; flags=$2
mkdir -p "$R/dbg/obj_$tag"
objs=""
for u in sweep sweep_bf16 wgrad misc sample capudf api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$R/diffudf_amd/csrc/dudf_$u.hip" -o "$R/dbg/obj_$tag/$u.o" &
  objs="$objs $R/dbg/obj_$tag/$u.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/dbg/libdudf_$tag.so" $objs && rm -rf "$R/dbg/obj_$tag"
