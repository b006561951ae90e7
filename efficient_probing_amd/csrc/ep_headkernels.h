// Small kernels shared by the heads with a query chain, a per-row LayerNorm and a GELU MLP behind the token passes
// (SigLIP / V-JEPA: ep_siglip.hip, CaiT: ep_cait.hip).  `static`: every translation unit launches its own copy (no
// device linking).
#pragma once
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

// Column work over many rows: 1024-thread workgroups = 32 column lanes x 32 row lanes (a row lane walks rows ty, ty + 32, ...);
// the 32 row-lane partials of a column are summed in a fixed order.
__device__ __forceinline__ float red32(float v, float (*sm)[33], int tx, int ty) {
  __syncthreads();
  sm[ty][tx] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) t += sm[i][tx];
  return t;
}

// q[j] = Wq[j,:] . latent + bq[j]      (one wave per output)
static __global__ __launch_bounds__(256) void ep_siglip_q_kernel(const float* __restrict__ latent, const float* __restrict__ Wq,
                                                        const float* __restrict__ bq, int D, float* __restrict__ q) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(Wq[(int64_t)j * D + d], latent[d], acc);
  acc = wave_sum(acc);
  if (lane == 0) q[j] = acc + bq[j];
}

// u[h,d] = scale * sum_c q[h*dh + c] * Wk[h*dh + c, d]        (grid (D / 32, H), 1024 threads)
static __global__ __launch_bounds__(1024) void ep_siglip_u_kernel(const float* __restrict__ q, const float* __restrict__ Wk, int D,
                                                         int dh, float scale, float* __restrict__ u) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx, h = blockIdx.y;
  float acc = 0.f;
  if (d < D)
    for (int c = ty; c < dh; c += 32) acc = fmaf(q[h * dh + c], Wk[(int64_t)(h * dh + c) * D + d], acc);
  acc = red32(acc, sm, tx, ty);
  if (ty == 0 && d < D) u[(int64_t)h * D + d] = acc * scale;
}
static inline int siglip_u(const float* q, const float* Wk, int D, int H, int dh, float scale, float* u, hipStream_t st) {
  hipLaunchKernelGGL(ep_siglip_u_kernel, dim3((D + 31) / 32, H), dim3(1024), 0, st, q, Wk, D, dh, scale, u);
  EP_LAUNCH_CHECK("ep_siglip_u_kernel");
  return 0;
}

// dq[j] = scale * Wk[j,:] . du[h(j),:]   (one wave per output; also d q.bias)
static __global__ __launch_bounds__(256) void ep_siglip_dq_kernel(const float* __restrict__ du, const float* __restrict__ Wk, int D,
                                                         int dh, float scale, int accumulate, float* __restrict__ dq,
                                                         float* __restrict__ dbq) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D) return;
  const int h = j / dh, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(Wk[(int64_t)j * D + d], du[(int64_t)h * D + d], acc);
  acc = wave_sum(acc) * scale;
  if (lane == 0) { dq[j] = acc; dbq[j] = accumulate ? dbq[j] + acc : acc; }
}

// per 64-column block of d:  dWk[j,d] (+)= scale q[j] du[h(j),d];  dWq[j,d] (+)= dq[j] latent[d];
// dlatent[d] (+)= sum_j Wq[j,d] dq[j];  d kv.bias[:D] <- 0
// Two launches: the outer products (grid (D / 64, 16): 64 columns x 4 row lanes over a sixteenth of the rows) and the
// (1 x D) . (D x D) product for d latent (grid D / 32, 1024 threads).
static __global__ __launch_bounds__(256) void ep_siglip_qgrad_kernel(const float* __restrict__ q, const float* __restrict__ dq,
                                                            const float* __restrict__ du, const float* __restrict__ latent,
                                                            int D, int dh, float scale, int accumulate,
                                                            float* __restrict__ dWk, float* __restrict__ dWq,
                                                            float* __restrict__ dbk) {
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int d = blockIdx.x * 64 + tx;
  if (d >= D) return;
  const int per = (D + gridDim.y - 1) / gridDim.y;
  const int j0 = blockIdx.y * per, j1 = (j0 + per) < D ? (j0 + per) : D;
  const float ld = latent[d];
  for (int j = j0 + ty; j < j1; j += 4) {
    const float gk = scale * q[j] * du[(int64_t)(j / dh) * D + d];
    float* ok_ = dWk + (int64_t)j * D + d;
    *ok_ = accumulate ? *ok_ + gk : gk;
    const float gq = dq[j] * ld;
    float* oq = dWq + (int64_t)j * D + d;
    *oq = accumulate ? *oq + gq : gq;
  }
  if (!accumulate && ty == 0 && blockIdx.y == 0) dbk[d] = 0.f;
}
// dlatent[d] (+)= sum_j Wq[j,d] dq[j]
static __global__ __launch_bounds__(1024) void ep_siglip_dlatent_kernel(const float* __restrict__ dq, const float* __restrict__ Wq,
                                                               int D, int accumulate, float* __restrict__ dlatent) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (d < D)
    for (int j = ty; j < D; j += 32) acc = fmaf(Wq[(int64_t)j * D + d], dq[j], acc);
  acc = red32(acc, sm, tx, ty);
  if (ty == 0 && d < D) dlatent[d] = accumulate ? dlatent[d] + acc : acc;
}
static inline int siglip_qgrad(const float* q, const float* dq, const float* du, const float* latent, const float* Wq, int D, int dh,
                               float scale, int accumulate, float* dWk, float* dWq, float* dlatent, float* dbk, hipStream_t st) {
  hipLaunchKernelGGL(ep_siglip_qgrad_kernel, dim3((D + 63) / 64, 16), dim3(256), 0, st, q, dq, du, latent, D, dh, scale, accumulate,
                     dWk, dWq, dbk);
  hipLaunchKernelGGL(ep_siglip_dlatent_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, dq, Wq, D, accumulate, dlatent);
  EP_LAUNCH_CHECK("ep_siglip_qgrad kernels");
  return 0;
}

// exact GELU (nn.GELU default, erf form): h = gelu(pre)
static __global__ __launch_bounds__(256) void ep_gelu_kernel(const float* __restrict__ pre, int64_t n4, float* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f4 v = reinterpret_cast<const f4*>(pre)[i];
  f4 r;
  r.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f)); r.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
  r.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f)); r.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
  reinterpret_cast<f4*>(h)[i] = r;
}
// g <- g * gelu'(pre),  gelu'(x) = Phi(x) + x phi(x)
static __global__ __launch_bounds__(256) void ep_gelu_bwd_kernel(const float* __restrict__ pre, int64_t n4, float* __restrict__ g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f4 v = reinterpret_cast<const f4*>(pre)[i];
  f4 d = reinterpret_cast<f4*>(g)[i];
  auto dg = [](float x) { return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x); };
  d.x *= dg(v.x); d.y *= dg(v.y); d.z *= dg(v.z); d.w *= dg(v.w);
  reinterpret_cast<f4*>(g)[i] = d;
}

static __global__ __launch_bounds__(256) void ep_rowscale_kernel(const float* __restrict__ u, const float* __restrict__ g, int rows,
                                                        int D, float* __restrict__ w) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * D) w[i] = u[i] * g[i % D];
}
static __global__ __launch_bounds__(256) void ep_vecadd_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                      float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
// h[b,:] = (x[b,:] - mean_b) rstd_b * g + beta        (stats (B,2) from the token-statistics kernel with N = 1)
static __global__ __launch_bounds__(256) void ep_rowln_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ g, const float* __restrict__ beta,
                                                           int64_t n, int D, float* __restrict__ h) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / D; const int d = (int)(i % D);
  h[i] = fmaf((x[i] - stats[2 * b]) * stats[2 * b + 1], g[d], beta[d]);
}
// LayerNorm backward per row, fused with the residual: dx[b,:] = res[b,:] + rstd (gd - mean(gd) - xhat mean(gd xhat)),
// gd = dh * g   (one wave per row)
static __global__ __launch_bounds__(256) void ep_rowln_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                         const float* __restrict__ stats, const float* __restrict__ g,
                                                         const float* __restrict__ res, int B, int D, float* __restrict__ dx) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const float mu = stats[2 * b], r = stats[2 * b + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float gd = dh[(int64_t)b * D + d] * g[d];
    s1 += gd; s2 = fmaf(gd, (x[(int64_t)b * D + d] - mu) * r, s2);
  }
  const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
  for (int d = lane; d < D; d += 64) {
    const float gd = dh[(int64_t)b * D + d] * g[d];
    const float xh = (x[(int64_t)b * D + d] - mu) * r;
    dx[(int64_t)b * D + d] = (res ? res[(int64_t)b * D + d] : 0.f) + r * (gd - m1 - xh * m2);
  }
}
// d g[d] (+)= sum_b dh[b,d] xhat[b,d];  d beta[d] (+)= sum_b dh[b,d]     (64 columns per workgroup, 4 row lanes)
static __global__ __launch_bounds__(1024) void ep_lnaffine_grad_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                              const float* __restrict__ stats, int B, int D, int accumulate,
                                                              float* __restrict__ dg, float* __restrict__ dbeta) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx;
  float ag = 0.f, ab = 0.f;
  if (d < D)
    for (int b = ty; b < B; b += 32) {
      const float v = dh[(int64_t)b * D + d];
      ag = fmaf(v, (x[(int64_t)b * D + d] - stats[2 * b]) * stats[2 * b + 1], ag); ab += v;
    }
  const float sg = red32(ag, sm, tx, ty), sb = red32(ab, sm, tx, ty);
  if (ty == 0 && d < D) {
    dg[d] = accumulate ? dg[d] + sg : sg;
    dbeta[d] = accumulate ? dbeta[d] + sb : sb;
  }
}
static inline int lnaffine_grad(const float* dh, const float* x, const float* stats, int B, int D, int accumulate, float* dg,
                                float* dbeta, hipStream_t st) {
  hipLaunchKernelGGL(ep_lnaffine_grad_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, dh, x, stats, B, D, accumulate, dg, dbeta);
  EP_LAUNCH_CHECK("ep_lnaffine_grad_kernel");
  return 0;
}


}  // namespace ep
