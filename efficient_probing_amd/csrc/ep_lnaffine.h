// Small kernels shared by the heads whose keys / values are LayerNorms of the tokens (CAE: ep_coca.hip, JEPA:
// ep_siglip.hip): folding the LayerNorm's affine part into the query rows and the value projection, and the
// gradients of that folding.  `static`: every translation unit launches its own copy (no device linking).
#pragma once
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

// Wv'[r,d] = Wv[r,d] gv[d];  bo[r] = Wv[r,:] . bv      (one wave per row)
static __global__ __launch_bounds__(256) void ep_cae_wv_kernel(const float* __restrict__ Wv, const float* __restrict__ gv,
                                                             const float* __restrict__ bv, int D, float* __restrict__ Wvs,
                                                             float* __restrict__ bo, const float* __restrict__ badd) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= D) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float wv = Wv[(int64_t)r * D + d];
    Wvs[(int64_t)r * D + d] = wv * gv[d];
    acc = fmaf(wv, bv[d], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) bo[r] = acc + (badd ? badd[r] : 0.f);
}

// du[h,d] = gk[d] dw[h,d];  d gk[d] (+)= sum_h u[h,d] dw[h,d];  d bk[d] <- 0 (the key-side shift cancels in the softmax)
static __global__ __launch_bounds__(256) void ep_cae_du_kernel(const float* __restrict__ dw, const float* __restrict__ u,
                                                      const float* __restrict__ gk, int D, int H, int accumulate,
                                                      float* __restrict__ du, float* __restrict__ dgk, float* __restrict__ dbk) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float g = 0.f;
  for (int h = 0; h < H; ++h) {
    const float v = dw[(int64_t)h * D + d];
    du[(int64_t)h * D + d] = gk[d] * v;
    g = fmaf(u[(int64_t)h * D + d], v, g);
  }
  dgk[d] = accumulate ? dgk[d] + g : g;
  if (!accumulate) dbk[d] = 0.f;
}

// value side, per 32-column block of d (1024 threads = 32 column lanes x 32 row lanes; launch with grid (D + 31) / 32): dWv[r,d] (+)= dWvs[r,d] gv[d] + dbo[r] bv[d];
// d gv[d] (+)= sum_r dWvs[r,d] Wv[r,d];  d bv[d] (+)= sum_r dbo[r] Wv[r,d];  unused norm2_cross gradients <- 0
static __global__ __launch_bounds__(1024) void ep_cae_dwv_kernel(const float* __restrict__ dWvs, const float* __restrict__ dbo,
                                                        const float* __restrict__ Wv, const float* __restrict__ gv,
                                                        const float* __restrict__ bv, int D, int accumulate,
                                                        float* __restrict__ dWv, float* __restrict__ dgv, float* __restrict__ dbv,
                                                        float* __restrict__ dn2w, float* __restrict__ dn2b) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 column lanes x 32 row lanes
  const int d = blockIdx.x * 32 + tx;
  const bool ok = d < D;
  float ag = 0.f, ab = 0.f;
  if (ok) {
    const float g = gv[d], b = bv[d];
    for (int r = ty; r < D; r += 32) {
      const float ds = dWvs[(int64_t)r * D + d], wv = Wv[(int64_t)r * D + d], db = dbo[r];
      const float v = fmaf(ds, g, db * b);
      float* o = dWv + (int64_t)r * D + d;
      *o = accumulate ? *o + v : v;
      ag = fmaf(ds, wv, ag); ab = fmaf(db, wv, ab);
    }
  }
  __syncthreads(); sm[ty][tx] = ag; __syncthreads();
  float sg = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) sg += sm[i][tx];
  __syncthreads(); sm[ty][tx] = ab; __syncthreads();
  float sb = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) sb += sm[i][tx];
  if (ty == 0 && ok) {
    dgv[d] = accumulate ? dgv[d] + sg : sg;
    dbv[d] = accumulate ? dbv[d] + sb : sb;
    if (!accumulate && dn2w) { dn2w[d] = 0.f; dn2b[d] = 0.f; }
  }
}


}  // namespace ep
