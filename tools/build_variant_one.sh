#!/usr/bin/env bash
# Build an A/B variant of the library where ONE translation unit is compiled with extra flags and the others are the
# objects of the shipped build: tools/build_variant_one.sh TAG ep_pool_mm2 "-DEP_MM2_ABLATE=1"
# -> efficient_probing_amd/variants/libep_hip_TAG.so  (select with EP_HIP_LIB=...)
set -euo pipefail
tag="$1"; unit="$2"; flags="${3:-}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/efficient_probing_amd/csrc"; out="$root/efficient_probing_amd/variants"; tmp="/tmp/epvar1_$tag"
mkdir -p "$out" "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $flags -c "$src/$unit.hip" -o "$tmp/$unit.o"
objs=()
for o in "$src"/*.o; do
  if [[ "$(basename "$o")" == "$unit.o" ]]; then objs+=("$tmp/$unit.o"); else objs+=("$o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libep_hip_$tag.so" "${objs[@]}"
echo "$out/libep_hip_$tag.so"
