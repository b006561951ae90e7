// dP[b, q, :] = sum_{c < DQ} dy[b, q DQ + c] Wv[q DQ + c, :]  for THIN query slices (DQ = D' / Q <= 64) on the bf16 matrix cores
// at fp32 accuracy -- the autograd of the value projection (reference poolings/ep.py:40) at the published protocol's 32 queries
// (main_linprobe.py:113: slices of 24 / 32 / 36 columns at D = 768 / 1024 / 1152).
//
// Round 5 ran it on the vector ALU (ep_tail.hip: ep_dp_thin_kernel, weights in registers, one scalar-broadcast FMA row per
// image): 54 us at 1024 x 32 x 768 and 66 us at 196 x 1024 (rocprofv3, in the step) for 100 / 134 MB of output -- 2.5 x the
// time the stores alone need.  Here one matrix instruction covers a whole slice: K = 32 >= DQ -- two (template KS = 2) for slices
// of 36 .. 64 columns: SigLIP2 SO400M's 1152 / 32 = 36, which ran on the exact-f32 LDS-DMA kernel at 145 us in the step (rocprofv3,
// profiles/r06/c4_q32_step_timeline.txt; K = 36 is below the bf16 tile's floor).
//
//   C'[d][b] = sum_k Wv^T[d][k] dy^T[k][b]     A' = the slice of Wv, transposed (rows d, k = the slice's rows, zero padded to 32)
//                                              B' = dy rows (k contiguous in memory: two 16-byte loads per lane, masked at DQ)
// so a lane of the result holds FOUR CONSECUTIVE d of one image: one 16-byte store per lane and 16 x 16 block, the four
// lane groups of a block write 64 contiguous bytes per image and the next block of the same wave completes the 128-byte line.
// A workgroup = (query, chunk of DCH output columns, RPW images): it splits its DQ x DCH piece of Wv once -- three bf16 terms,
// transposed into the plane images of ep_wgrad3.h (same staging code, same conflict-free fragment reads) -- then every wave
// walks 16-image blocks: split its dy fragment (44 vector instructions), and per 16-column block three fragment reads, six
// matrix instructions (one in the AMP-bf16 mode) and one store.  Bound: the output stores.
#include "ep_wgrad3.h"

namespace ep {

constexpr int DPS_ROWS = 256;                       // images per workgroup (four 64-image rounds of its four waves)
constexpr int DPS_PITCH = 272;                      // bytes per image row of the wave-private transpose tile (64 floats + 16: conflict-free)
constexpr int DPS_TRW = 16 * DPS_PITCH;             // ... per wave

template <int NSUB, int KS = 1>                     // 64-column sub-tiles per workgroup: DCH = 64 NSUB; KS = matrix K-steps of 32 slice rows
__global__ __launch_bounds__(256) void ep_dp_slice_kernel(const float* __restrict__ dy, const float* __restrict__ Wv, int B, int D, int Dp,
                                                         int Q, int DQ, int one, float* __restrict__ dP,
                                                         const float* __restrict__ yv, float* __restrict__ ML) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;
  const int q = blockIdx.x, d0 = blockIdx.y * (64 * NSUB), b00 = blockIdx.z * DPS_ROWS;
  // ---- the slice of Wv, rows q DQ .. + DQ, columns d0 .. + 64 NSUB: split and transposed into plane images [sub][term][d][k] ----
  {
    const int mq = tid & 15, kp = tid >> 4;
    const float* Ws = Wv + (int64_t)q * DQ * D;
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f4v x[2];
        w3_load_T(Ws, D, D, DQ, d0 + 64 * s, 32 * ks, kp, mq, x);
        w3_stage_T(lds + (s * KS + ks) * 3 * W3_IMG, x, D, DQ, d0 + 64 * s, 32 * ks, kp, mq, one != 0);
      }
  }
  __syncthreads();
  const float* dyq = dy + (int64_t)q * DQ;
  for (int r = 0; r < DPS_ROWS / 64; ++r) {
    const int b = b00 + r * 64 + w * 16 + i16;
    if (b00 + r * 64 + w * 16 >= B) break;           // (wave-uniform)
    // this lane's eight dy values: image b, slice columns 8 kk .. 8 kk + 7 (zero at and beyond DQ: the next query's columns)
    const float* src = dyq + (int64_t)(b < B ? b : B - 1) * Dp;
    float vk[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k0 = 32 * ks + 8 * kk;
      const f4v v0 = *reinterpret_cast<const f4v*>(src + (k0 < DQ ? k0 : 0));
      const f4v v1 = *reinterpret_cast<const f4v*>(src + (k0 + 4 < DQ ? k0 + 4 : 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) { vk[ks][e] = (k0 + e < DQ) ? v0[e] : 0.f; vk[ks][4 + e] = (k0 + 4 + e < DQ) ? v1[e] : 0.f; }
    }
    // the softmax-correction rows delta[b, q] = dy[b, q-slice] . y[b, q-slice] (ep_delta_kernel's sum, one launch less): the
    // workgroups of the first column chunk hold exactly these dy values -- eight products per lane, the four lane groups of an
    // image combined in fixed order
    if (ML && blockIdx.y == 0) {
      const float* ys = yv + (int64_t)q * DQ + (int64_t)(b < B ? b : B - 1) * Dp;
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 32 * ks + 8 * kk;
        const f4v y0 = *reinterpret_cast<const f4v*>(ys + (k0 < DQ ? k0 : 0));
        const f4v y1 = *reinterpret_cast<const f4v*>(ys + (k0 + 4 < DQ ? k0 + 4 : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc = fmaf(vk[ks][e], y0[e], acc); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc = fmaf(vk[ks][4 + e], y1[e], acc); }     // (v is zero at and beyond DQ)
      }
      const float a1 = __shfl_xor(acc, 16, 64);
      const float s01 = acc + a1;                      // groups (0, 1) and (2, 3)
      const float s23 = __shfl_xor(s01, 32, 64);
      if (kk == 0 && b < B) ML[((int64_t)b * Q + q) * 4 + 2] = s01 + s23;
    }
    pl_u4 bt[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pl_split8(vk[ks], bt[ks]);
    // The 16 x 64 result of a sub-tile goes through a wave-private LDS tile (rows = images, 272-byte pitch) so that the global
    // stores are WHOLE 256-byte row segments -- four images per instruction -- instead of 64-byte pieces of sixteen images
    // 98 KiB apart (first form of this kernel: 54 us, as slow as the vector-ALU one; the stores are the bound).
    char* tr = lds + NSUB * KS * 3 * W3_IMG + w * DPS_TRW;
    const int bw = b00 + r * 64 + w * 16;            // first image of this wave's block
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        const int off = w3_off(16 * bi + i16, kk);
        f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const char* img = lds + (s * KS + ks) * 3 * W3_IMG;
          if (one) {
            const pl_u4 a0 = *reinterpret_cast<const pl_u4*>(img + off);
            acc = pl_mfma(a0, bt[ks][0], acc);
          } else {
            const pl_u4 a0 = *reinterpret_cast<const pl_u4*>(img + off);
            const pl_u4 a1 = *reinterpret_cast<const pl_u4*>(img + W3_IMG + off);
            const pl_u4 a2 = *reinterpret_cast<const pl_u4*>(img + 2 * W3_IMG + off);
            // smallest terms first (as ep_planes.hip): lo x hi, hi x lo, mid x mid, mid x hi, hi x mid, hi x hi
            acc = pl_mfma(a2, bt[ks][0], acc);
            acc = pl_mfma(a0, bt[ks][2], acc);
            acc = pl_mfma(a1, bt[ks][1], acc);
            acc = pl_mfma(a1, bt[ks][0], acc);
            acc = pl_mfma(a0, bt[ks][1], acc);
            acc = pl_mfma(a0, bt[ks][0], acc);
          }
        }
        *reinterpret_cast<f4v*>(tr + i16 * DPS_PITCH + (16 * bi + 4 * kk) * 4) = acc;      // image i16, columns 16 bi + 4 kk ..
      }
      // (wave-private: the reads below only need this wave's own writes -- the compiler's lgkmcnt wait in front of them)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int im = 4 * it + kk;                   // image of the block, 16 lanes x 16 bytes = its 64 columns
        const f4v o = *reinterpret_cast<const f4v*>(tr + im * DPS_PITCH + i16 * 16);
        if (bw + im < B && d0 + 64 * s + 4 * i16 < D)
          *reinterpret_cast<f4v*>(dP + ((int64_t)(bw + im) * Q + q) * D + d0 + 64 * s + 4 * i16) = o;
      }
    }
  }
}

// EP_DP_SLICE=0: the vector-ALU kernel of round 5 (ep_tail.hip)
bool project_dp_slice_ok(const float* dy, const float* Wv, const float* dP, int D, int Dp, int Q) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_DP_SLICE"); on = e ? atoi(e) : 1; }
  const int Dq = Q > 0 ? Dp / Q : 0;
  return on && Q > 0 && Dp % Q == 0 && Dq >= 4 && Dq <= 64 && Dq % 4 == 0 && D % 64 == 0 && Dp % 4 == 0 && aligned16(dy) && aligned16(Wv) && aligned16(dP);
}

// yv / ML (both or neither): also write delta[b, q] = dy[b, q-slice] . yv[b, q-slice] to ML[b, q, 2]
int project_dp_slice(const float* dy, const float* Wv, int B, int D, int Dp, int Q, float* dP, hipStream_t st, const float* yv, float* ML) {
  const int Dq = Dp / Q;
  const int one = (gemm_arith() == 1) ? 1 : 0;
  const int ks = Dq > 32 ? 2 : 1;
  // (two K-steps double the plane images: at most two sub-tiles then)
  const int nsub = (D % 256 == 0 && ks == 1) ? 4 : D % 128 == 0 ? 2 : 1;
  const dim3 grid(Q, D / (64 * nsub), (B + DPS_ROWS - 1) / DPS_ROWS), block(256);
  const size_t lds = (size_t)nsub * ks * 3 * W3_IMG + 4 * DPS_TRW;
#define EP_DPS_LAUNCH(NS, KSV)                                                                                                      \
  do {                                                                                                                               \
    static bool attr = false;                                                                                                        \
    if (!attr) { (void)hipFuncSetAttribute((const void*)ep_dp_slice_kernel<NS, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; } \
    hipLaunchKernelGGL((ep_dp_slice_kernel<NS, KSV>), grid, block, lds, st, dy, Wv, B, D, Dp, Q, Dq, one, dP, yv, ML);               \
  } while (0)
  if (ks == 2) { if (nsub == 2) EP_DPS_LAUNCH(2, 2); else EP_DPS_LAUNCH(1, 2); }
  else if (nsub == 4) EP_DPS_LAUNCH(4, 1);
  else if (nsub == 2) EP_DPS_LAUNCH(2, 1);
  else EP_DPS_LAUNCH(1, 1);
#undef EP_DPS_LAUNCH
  EP_LAUNCH_CHECK("ep_dp_slice_kernel");
  return 0;
}

}  // namespace ep
