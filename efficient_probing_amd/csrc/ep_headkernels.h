// Small kernels shared by the heads with a query chain, a per-row LayerNorm and a GELU MLP behind the token passes
// (SigLIP / V-JEPA: ep_siglip.hip, CaiT: ep_cait.hip).  `static`: every translation unit launches its own copy (no
// device linking).
#pragma once
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

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

// u[h,d] = scale * sum_c q[h*dh + c] * Wk[h*dh + c, d]
static __global__ __launch_bounds__(256) void ep_siglip_u_kernel(const float* __restrict__ q, const float* __restrict__ Wk, int D,
                                                        int dh, float scale, float* __restrict__ u) {
  const int d = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (d >= D) return;
  float acc = 0.f;
  for (int c = 0; c < dh; ++c) acc = fmaf(q[h * dh + c], Wk[(int64_t)(h * dh + c) * D + d], acc);
  u[(int64_t)h * D + d] = acc * scale;
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
static __global__ __launch_bounds__(256) void ep_siglip_qgrad_kernel(const float* __restrict__ q, const float* __restrict__ dq,
                                                            const float* __restrict__ du, const float* __restrict__ latent,
                                                            const float* __restrict__ Wq, int D, int dh, float scale,
                                                            int accumulate, float* __restrict__ dWk, float* __restrict__ dWq,
                                                            float* __restrict__ dlatent, float* __restrict__ dbk) {
  extern __shared__ float sh[];          // q[D] | dq[D] | partial[4][64]
  float* s_q = sh; float* s_dq = sh + D; float* part = sh + 2 * D;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  for (int i = tid; i < D; i += 256) { s_q[i] = q[i]; s_dq[i] = dq[i]; }
  __syncthreads();
  const int d = blockIdx.x * 64 + tx;
  const bool ok = d < D;
  float acc = 0.f;
  if (ok) {
    const float ld = latent[d];
    for (int j = ty; j < D; j += 4) {
      const float gk = scale * s_q[j] * du[(int64_t)(j / dh) * D + d];
      float* ok_ = dWk + (int64_t)j * D + d;
      *ok_ = accumulate ? *ok_ + gk : gk;
      const float gq = s_dq[j] * ld;
      float* oq = dWq + (int64_t)j * D + d;
      *oq = accumulate ? *oq + gq : gq;
      acc = fmaf(Wq[(int64_t)j * D + d], s_dq[j], acc);
    }
    if (!accumulate && ty == 0) dbk[d] = 0.f;
  }
  part[ty * 64 + tx] = acc;
  __syncthreads();
  if (ty == 0 && ok) {
    const float g = (part[tx] + part[64 + tx]) + (part[128 + tx] + part[192 + tx]);
    dlatent[d] = accumulate ? dlatent[d] + g : g;
  }
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
static __global__ __launch_bounds__(256) void ep_lnaffine_grad_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                             const float* __restrict__ stats, int B, int D, int accumulate,
                                                             float* __restrict__ dg, float* __restrict__ dbeta) {
  __shared__ float pg[4][64], pb[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + tx;
  float ag = 0.f, ab = 0.f;
  if (d < D)
    for (int b = ty; b < B; b += 4) {
      const float v = dh[(int64_t)b * D + d];
      ag = fmaf(v, (x[(int64_t)b * D + d] - stats[2 * b]) * stats[2 * b + 1], ag); ab += v;
    }
  pg[ty][tx] = ag; pb[ty][tx] = ab;
  __syncthreads();
  if (ty == 0 && d < D) {
    const float sg = (pg[0][tx] + pg[1][tx]) + (pg[2][tx] + pg[3][tx]), sb = (pb[0][tx] + pb[1][tx]) + (pb[2][tx] + pb[3][tx]);
    dg[d] = accumulate ? dg[d] + sg : sg;
    dbeta[d] = accumulate ? dbeta[d] + sb : sb;
  }
}


}  // namespace ep
