// Side tasks of the second token pass (the weight-gradient contractions, the bias column sum and the statistics fold run
// as extra workgroups of its launch): the dispatcher over the task list.  Included by the token-pass kernels only.
#pragma once
#include "ep_side.h"
#include "ep_wgrad3.h"

namespace ep {

// ---- side tasks of the second token pass ---------------------------------------------------------
// Workgroups with blockIdx.x >= (pooling workgroups) run these instead of streaming tokens.  They are
// dispatched as pooling workgroups retire, i.e. into the tail of the pass where the chip would
// otherwise drain; none of them feeds anything before the optimizer.  All T/T-layout vector GEMMs.
constexpr size_t SIDE_LDS_BYTES = sizeof(float) * 2 * 2 * LDS_OPERAND > W3_LDS_BYTES ? sizeof(float) * 2 * 2 * LDS_OPERAND : W3_LDS_BYTES;

// DMA_OK: the launch has the LDS of the ring form (w3d_lds_bytes) and may run SideTasks.b3 == 2
template <bool DMA_OK = false>
__device__ __forceinline__ void run_side_task(const SideTasks& s, int t, char* lds_raw) {
  auto lds = reinterpret_cast<float (*)[2][LDS_OPERAND]>(lds_raw);
  const int tglob = t;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int n = s.gx[i] * s.gy[i] * s.gz[i];
    if (t < n) {
      // XCD-aware order: workgroups go to the 8 XCDs round robin, each XCD has its own L2, and the tiles of one
      // contraction re-read each other's operand panels.  Tile index u = (XCD) * n/8 + (position on that XCD): every XCD
      // works on a contiguous range of the natural (bx fastest, then by, then batch) order -- two row blocks of dWc, one
      // query of dWv -- so a panel is fetched into ONE L2 instead of eight while the token stream flushes them all.
      if (n % 8 == 0 && (tglob - t) % 8 == 0 && s.xcd_order) t = (t % 8) * (n / 8) + t / 8;
      const int bx = t % s.gx[i], r = t / s.gx[i];
      const int by = r % s.gy[i], bz = r / s.gy[i];
      // (round 3: the LDS-DMA tile body of ep_gemm_dma.h as the side body -- 3 stages, 4 symmetric waves, 64-row tiles --
      // measured SLOWER inside the pass: second pass 181 -> 192 us at 256x768, 155 -> 166 us at 197x768; removed again)
      if constexpr (DMA_OK) {
        if (s.b3 == 2) {                               // ... with the rows prefetched into an LDS ring (few tiles per CU: ep_wgrad3.h)
          if (s.bm[i] == 64) gemm_tile_b3d<64>(s.g[i], bx, by, bz, lds_raw);
          else gemm_tile_b3d<32>(s.g[i], bx, by, bz, lds_raw);
          return;
        }
      }
      if (s.b3) {                                      // bf16 x3 at fp32 accuracy: 96 instead of 256 matrix cycles per block
        if (s.bm[i] == 64) gemm_tile_b3<64>(s.g[i], bx, by, bz, lds_raw);
        else gemm_tile_b3<32>(s.g[i], bx, by, bz, lds_raw);
        return;
      }
      if (s.bm[i] == 64) gemm_tile<false, false, true, 64>(s.g[i], bx, by, bz, lds);
      else gemm_tile<false, false, true, 32>(s.g[i], bx, by, bz, lds);
      return;
    }
    t -= n;
  }
  if (t < s.n_colsum) {
    colsum_block(s.cs_src, s.cs_B, s.cs_ncol, s.cs_ld, s.cs_accumulate, s.cs_out, t,
                 reinterpret_cast<float (*)[CG]>(lds_raw));
    return;
  }
  t -= s.n_colsum;
  if (t < s.n_stats) { ce_stats_block(s.rowstat, s.rs_B, s.stats, reinterpret_cast<f4*>(lds_raw)); return; }
  t -= s.n_stats;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < s.n_xcs) {
      if (t < s.xcs_blocks[i]) {
        colsum_block(s.xcs_src[i], s.xcs_B[i], s.xcs_ncol[i], s.xcs_ld[i], s.xcs_acc[i], s.xcs_out[i], t,
                     reinterpret_cast<float (*)[CG]>(lds_raw));
        return;
      }
      t -= s.xcs_blocks[i];
    }
  }
}

}  // namespace ep
