#!/usr/bin/env bash
# Build an A/B variant of the library: tools/build_variant.sh TAG "-DEP_STREAM_TT=8 ..."
# -> efficient_probing_amd/variants/libep_hip_TAG.so  (select with EP_HIP_LIB=...)
set -euo pipefail
tag="$1"; flags="${2:-}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/efficient_probing_amd/csrc"; out="$root/efficient_probing_amd/variants"; tmp="/tmp/epvar_$tag"
mkdir -p "$out" "$tmp"
for f in ep_pool ep_pool_stream ep_pool_bwd2 ep_pool_mfma ep_pool_mm ep_pool_mb ep_pool_wide ep_gemm ep_planes ep_tail ep_optim ep_api ep_coca ep_abmilp ep_knn ep_siglip ep_aim ep_pool_imgq ep_simpool ep_cait ep_clip ep_dolg ep_cbam ep_dinovit; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $flags -c "$src/$f.hip" -o "$tmp/$f.o" 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libep_hip_$tag.so" "$tmp"/*.o
echo "$out/libep_hip_$tag.so"
