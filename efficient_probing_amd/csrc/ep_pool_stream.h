// Compile-time geometry of the streaming pool kernels, shared by the kernels (template
// parameters) and the host-side planner so that both always agree.
#pragma once
#include "ep_internal.h"

namespace ep {

constexpr int STREAM_TT = 4;                       // tokens per ring tile

constexpr int stream_kdma(int kp, int nw) { return (STREAM_TT * kp + nw - 1) / nw; }
constexpr int stream_nslot(int kp, int nw, bool bwd) {
  const int slot = STREAM_TT * kp * 1024 + (bwd ? nw * 256 : 0);   // worst case D = 256*kp
  const int budget = 160 * 1024 / (8 / nw);                        // 8 waves per CU
  int ns = budget / slot;
  if (ns > 8) ns = 8;
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
int stream_launch(bool bwd, const StreamPlan& c, const PoolParams& p, hipStream_t st);

}  // namespace ep
