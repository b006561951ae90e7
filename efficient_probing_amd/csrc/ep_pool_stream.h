// Compile-time geometry of the streaming pool kernels, shared by the kernels (template
// parameters) and the host-side planner so that both always agree.
#pragma once
#include "ep_internal.h"

namespace ep {

// build-time tuning knobs (A/B variants are built with -D...; defaults are the shipped values)
#ifndef EP_STREAM_TT
#define EP_STREAM_TT 8
#endif
#ifndef EP_STREAM_NSLOT_CAP
#define EP_STREAM_NSLOT_CAP 8
#endif
#ifndef EP_STREAM_WAVES_PER_CU
#define EP_STREAM_WAVES_PER_CU 8
#endif
#ifndef EP_DMA_AUX
#define EP_DMA_AUX 2              // cache-policy bits of the LDS-DMA loads (2 = nt: streamed once)
#endif
constexpr int STREAM_TT = EP_STREAM_TT;            // tokens per ring tile
constexpr int STREAM_WAVES_PER_CU = EP_STREAM_WAVES_PER_CU;

constexpr int stream_kdma(int kp, int nw) { return (STREAM_TT * kp + nw - 1) / nw; }
constexpr int stream_nslot(int kp, int nw, bool bwd) {
  const int slot = STREAM_TT * kp * 1024 + (bwd ? nw * 256 : 0);   // worst case D = 256*kp
  const int budget = 160 * 1024 / (STREAM_WAVES_PER_CU / nw);      // resident workgroups share the LDS
  int ns = budget / slot;
  if (ns > EP_STREAM_NSLOT_CAP) ns = EP_STREAM_NSLOT_CAP;
  const int kd = stream_kdma(kp, nw) + (bwd ? 1 : 0);
  while (ns > 3 && (ns - 2) * kd > 60) --ns;                       // vmcnt is a 6-bit counter
  return ns;
}
constexpr bool stream_valid(int qw, int kp, int nw) {
  return stream_nslot(kp, nw, false) >= 3 && stream_nslot(kp, nw, true) >= 3 &&
         (stream_nslot(kp, nw, true) - 2) * (stream_kdma(kp, nw) + 1) <= 60 &&
         !(qw == 4 && kp > 3) && !(qw == 2 && kp > 5);             // register budget (spills beyond)
}

template <int QW, int KP, int NW>
struct StreamCfgT {
  static constexpr int KDMA = stream_kdma(KP, NW);
  static constexpr int NSLOT_F = stream_nslot(KP, NW, false);
  static constexpr int NSLOT_B = stream_nslot(KP, NW, true);
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
int stream_launch(bool bwd, const StreamPlan& c, const PoolParams& p, hipStream_t st);

}  // namespace ep
