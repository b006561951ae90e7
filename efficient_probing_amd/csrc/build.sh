#!/usr/bin/env bash
# Build libep_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libep_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-variable"
objs=()
maxjobs="${EP_BUILD_JOBS:-8}"
for f in ep_pool ep_pool_stream ep_pool_bwd2 ep_pool_mfma ep_pool_mm ep_pool_mm2 ep_pool_mb ep_pool_mb_amp ep_pool_wide ep_pool_wideb ep_gemm ep_planes ep_planes_big ep_tail ep_dp_slice ep_optim ep_api ep_coca ep_abmilp ep_knn ep_siglip ep_aim ep_pool_imgq ep_simpool ep_cait ep_clip ep_dolg ep_cbam ep_dinovit; do
  src="${here}/${f}.hip"; obj="${here}/${f}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "${here}/ep_common.h" -nt "$obj" || "${here}/ep_internal.h" -nt "$obj" || "${here}/ep_pool_stream.h" -nt "$obj" || "${here}/ep_side.h" -nt "$obj" || "${here}/ep_inpass.h" -nt "$obj" || "${here}/ep_gemm_dma.h" -nt "$obj" || "${here}/ep_sidetask.h" -nt "$obj" || "${here}/ep_stream_dev.h" -nt "$obj" || "${here}/ep_lnaffine.h" -nt "$obj" || "${here}/ep_pool_imgq.h" -nt "$obj" || "${here}/ep_headkernels.h" -nt "$obj" || "${here}/ep_planes_dev.h" -nt "$obj" || "${here}/ep_wgrad3.h" -nt "$obj" || "${here}/../../include/ep_hip.h" -nt "$obj" || ( "$f" == ep_pool_mb_amp && "${here}/ep_pool_mb.hip" -nt "$obj" ) ]]; then
    echo "[build] hipcc ${f}.hip" >&2
    while (( $(jobs -rp | wc -l) >= maxjobs )); do wait -n || true; done
    rm -f "${obj}.failed"
    ( "$HIPCC" $FLAGS ${EP_EXTRA_FLAGS:-} -c "$src" -o "${obj}.tmp" && mv -f "${obj}.tmp" "$obj" || touch "${obj}.failed" ) &
  fi
  objs+=("$obj")
done
# a compile that fails (or is killed) must fail the build: never link a stale object
wait
fail=0
for o in "${objs[@]}"; do
  if [[ -f "${o}.failed" || ! -f "$o" ]]; then echo "[build] compiling ${o%.o}.hip failed" >&2; rm -f "${o}.failed" "${o}.tmp"; fail=1; fi
done
if (( fail )); then exit 1; fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "[build] $out" >&2
