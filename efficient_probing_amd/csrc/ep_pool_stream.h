// Compile-time geometry of the streaming pool kernels, shared by the kernels (template
// parameters) and the host-side planner so that both always agree.
#pragma once
#include "ep_internal.h"

namespace ep {

// build-time tuning knobs (A/B variants are built with -D...; defaults are the shipped values)
#ifndef EP_STREAM_NSLOT_CAP
#define EP_STREAM_NSLOT_CAP 8
#endif
#ifndef EP_STREAM_ABLATE
#define EP_STREAM_ABLATE 0        // diagnostic builds of the forward pass: 1 ring only (no arithmetic), 3 no pooling FMAs
#endif
#ifndef EP_DMA_AUX
#define EP_DMA_AUX 2              // cache-policy bits of the LDS-DMA loads (2 = nt: streamed once)
#endif

// Geometry per configuration (measured on MI355X, tools/ab_variants.sh):
//  * 2 queries per wave and D <= 768 (<= 168 VGPRs, compute-heavy): 12 waves per CU = three 4-wave
//    workgroups with rings of 4-token tiles -- the third workgroup hides the other two's barrier and
//    LDS latency (152 us vs 164 us per pass at 256x768, Q = 8);
//  * otherwise 8 waves per CU with 8-token tiles (fewer barriers per byte; with 1 query per wave the
//    kernel is memory-bound and the longer DMA bursts win: 0.76 vs 0.62 of peak at Q = 1).
constexpr int stream_waves_per_cu(int qw, int kp, int nw) { return (nw == 4 && qw == 2 && kp <= 3) ? 12 : 8; }
constexpr int stream_tt(int qw, int kp, int nw) {
  if (stream_waves_per_cu(qw, kp, nw) == 12) return 4;
  // 8-token tiles need a ring of >= 3 of them (backward: + the per-wave small pieces) in this workgroup's LDS share
  const int budget = 160 * 1024 / (stream_waves_per_cu(qw, kp, nw) / nw);
  return (3 * (8 * kp * 1024 + nw * 256) <= budget) ? 8 : 4;
}

constexpr int stream_kdma(int qw, int kp, int nw) { return (stream_tt(qw, kp, nw) * kp + nw - 1) / nw; }
constexpr int stream_nslot(int qw, int kp, int nw, bool bwd) {
  const int slot = stream_tt(qw, kp, nw) * kp * 1024 + (bwd ? nw * 256 : 0);   // worst case D = 256*kp
  const int budget = 160 * 1024 / (stream_waves_per_cu(qw, kp, nw) / nw);    // resident workgroups share the LDS
  int ns = budget / slot;
  if (ns > EP_STREAM_NSLOT_CAP) ns = EP_STREAM_NSLOT_CAP;
  const int kd = stream_kdma(qw, kp, nw) + (bwd ? 1 : 0);
  while (ns > 3 && (ns - 2) * kd > 60) --ns;                       // vmcnt is a 6-bit counter
  return ns;
}
constexpr bool stream_valid(int qw, int kp, int nw) {
  return stream_nslot(qw, kp, nw, false) >= 3 && stream_nslot(qw, kp, nw, true) >= 3 &&
         (stream_nslot(qw, kp, nw, true) - 2) * (stream_kdma(qw, kp, nw) + 1) <= 60 &&
         !(qw == 4 && kp > 3) && !(qw == 2 && kp > 5);             // register budget (spills beyond)
}

template <int QW, int KP, int NW>
struct StreamCfgT {
  static constexpr int TT = stream_tt(QW, KP, NW);                 // tokens per ring tile
  static constexpr int WPC = stream_waves_per_cu(QW, KP, NW);      // resident waves per CU
  static constexpr int KDMA = stream_kdma(QW, KP, NW);
  static constexpr int NSLOT_F = stream_nslot(QW, KP, NW, false);
  static constexpr int NSLOT_B = stream_nslot(QW, KP, NW, true);
  static constexpr bool VALID = stream_valid(QW, KP, NW);
};

struct StreamPlan {
  int qw, kp, nw, grid;
  bool ok;
};

StreamPlan stream_plan(int B, int N, int D, int Q);
// matrix-core variant (ep_pool_mfma.hip)
bool mf_supported(int D, int Q, int64_t cls_bstride);
int mf_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st);
// all-matrix-core variant (ep_pool_mm.hip)
bool mm_supported(int D, int Q, int64_t cls_bstride);
int mm_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st);
// ... with 17 .. 32 queries in one read of the tokens (ep_pool_mm2.hip): two 16-query blocks against the resident tile
bool mm2_supported(int D, int Q, int64_t cls_bstride, bool bwd);
int mm2_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st);
// bf16-token matrix-core variant (ep_pool_mb.hip): bf16-stored tokens, D in {256, 384, 512, 768, 1024, 1152}, Q <= 16
bool mb_supported(int D, int Q, int64_t cls_bstride);
int mb_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side = nullptr);
bool mb_takes_side(int D);                       // the second pass of this D can carry SideTasks
bool mb_takes_delta(int D, int Q, int Dv);
bool mb_takes_inpass_dp(const PoolParams& p);     // ... and run the dP contraction in front of its stream (ep_inpass.h)       // ... and compute the delta rows itself (PoolParams.dyv / yv / Dv)
int mb_grid(int D, int B);
// ... with 17 .. 32 queries in one read of the tokens (second half of ep_pool_mb.hip): D in {256, 384, 512, 768}
bool mbq_supported(int D, int Q, int64_t cls_bstride);
bool mbq_takes_delta(int D, int Q, int Dv);
int mbq_grid(int B);
int mbq_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st);
const char* mb_kernel_name(int D, bool bwd);
// wide-row variant (ep_pool_wide.hip): D = 2048 / 4096, Q <= 8, row split across the waves
bool wide_supported(int D, int Q, int64_t cls_bstride, int x_bf16 = 0);
int wide_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st);
int wide_grid(int D, int B, int x_bf16 = 0);
// ep_pool_wideb.hip: bf16-stored wide rows, scores on the matrix cores (the default; EP_POOL_WIDEB=0 switches it off)
bool wideb_supported(int D, int Q, int64_t cls_bstride, int x_bf16, bool bwd);
int wideb_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st);
// LayerNorm-of-tokens mode (PoolParams.tokstat) in the vector-ALU streaming kernels
bool stream_ln_supported(int D, int Q);
bool stream_ln_bf16_supported(int D, int Q);
int stream_launch(bool bwd, const StreamPlan& c, const PoolParams& p, hipStream_t st, const SideTasks* side = nullptr);
int stream_resident_blocks_per_cu(bool bwd, const StreamPlan& c, const PoolParams& p, const SideTasks* side = nullptr);

}  // namespace ep
