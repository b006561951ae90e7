// The small dense tail of the EP head: BatchNorm1d(affine=False) train/eval/backward
// (reference probe_heads.py:109-110), CrossEntropyLoss + top-k accuracy
// (reference main_linprobe.py:589, engine_finetune.py:62-63), bias gradient, and the softmax
// correction term delta of the pooling backward.  All fp32; batch reductions are done in a
// fixed order (no atomics on values that feed the parameters), so a step is reproducible.
#include "ep_side.h"

namespace ep {

// BatchNorm over the batch axis in two launches with whole-chip parallelism:
//   stage 1: grid (Dp/CG, RS): per row-chunk, per column: (mean_chunk, M2_chunk) [train] or
//            (sum dz, sum dz*z) [backward]  ->  partial[rs][which][col]
//   stage 2: same grid: every workgroup combines the RS partials of its columns in a FIXED order
//            (Chan's parallel variance formula -- as stable as the two-pass form), then normalises /
//            back-propagates its own row-chunk.  Deterministic, no atomics.
constexpr int RS_MAX = 32;

__device__ __forceinline__ void chunk_rows(int B, int nrs, int rs, int& r0, int& r1) {
  const int per = (B + nrs - 1) / nrs;
  r0 = rs * per;
  r1 = (r0 + per) < B ? (r0 + per) : B;
  if (r0 > B) r0 = B;
}

__global__ __launch_bounds__(256) void ep_bn_stats_kernel(const float* __restrict__ y, int B, int Dp,
                                                        float* __restrict__ partial) {
  __shared__ float sm[RL][CG];
  __shared__ float bc[CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  int r0, r1;
  chunk_rows(B, gridDim.y, blockIdx.y, r0, r1);
  const int n = r1 - r0;
  float s = 0.f;
  if (ok) for (int b = r0 + ty; b < r1; b += RL) s += y[(int64_t)b * Dp + col];
  s = colreduce(s, sm, tx, ty);
  if (ty == 0) bc[tx] = n > 0 ? s / (float)n : 0.f;
  __syncthreads();
  const float mu = bc[tx];
  float q = 0.f;
  if (ok) for (int b = r0 + ty; b < r1; b += RL) { const float d = y[(int64_t)b * Dp + col] - mu; q = fmaf(d, d, q); }
  q = colreduce(q, sm, tx, ty);
  if (ty == 0 && ok) {
    partial[((int64_t)blockIdx.y * 2 + 0) * Dp + col] = mu;
    partial[((int64_t)blockIdx.y * 2 + 1) * Dp + col] = q;
  }
}

__global__ __launch_bounds__(256) void ep_bn_apply_kernel(const float* __restrict__ y, int B, int Dp, float eps,
                                                        float momentum, const float* __restrict__ partial,
                                                        float* __restrict__ z, float* __restrict__ rstd_out,
                                                        float* __restrict__ rmean, float* __restrict__ rvar,
                                                        int64_t* __restrict__ nbt) {
  __shared__ float bc[2][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  const int nrs = gridDim.y;
  if (ty == 0 && ok) {
    float tot = 0.f;
    for (int r = 0; r < nrs; ++r) {
      int a0, a1; chunk_rows(B, nrs, r, a0, a1);
      tot += (float)(a1 - a0) * partial[((int64_t)r * 2 + 0) * Dp + col];
    }
    const float mu = tot / (float)B;
    float m2 = 0.f;
    for (int r = 0; r < nrs; ++r) {
      int a0, a1; chunk_rows(B, nrs, r, a0, a1);
      const float d = partial[((int64_t)r * 2 + 0) * Dp + col] - mu;
      m2 += partial[((int64_t)r * 2 + 1) * Dp + col] + (float)(a1 - a0) * d * d;
    }
    const float var = m2 / (float)B;                       // biased: used for normalisation
    const float rs = 1.0f / sqrtf(var + eps);
    bc[0][tx] = mu; bc[1][tx] = rs;
    if (blockIdx.y == 0) {
      rstd_out[col] = rs;
      const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
      rmean[col] = (1.0f - momentum) * rmean[col] + momentum * mu;
      rvar[col] = (1.0f - momentum) * rvar[col] + momentum * unbiased;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  __syncthreads();
  const float mu = bc[0][tx], rs = bc[1][tx];
  int r0, r1;
  chunk_rows(B, nrs, blockIdx.y, r0, r1);
  if (ok) for (int b = r0 + ty; b < r1; b += RL) z[(int64_t)b * Dp + col] = (y[(int64_t)b * Dp + col] - mu) * rs;
}

// Small batches (B <= 64 * BNF_RPT rows) in ONE launch: a workgroup of 1024 threads owns 16 columns and ALL rows -- 16 column
// lanes x 64 row lanes, each thread's rows in registers -- so the statistics, the normalisation (or its backward) and the
// running-statistics update need one read and one write of the (B, Dp) matrix and no partials.  Two-pass variance (mean, then
// centred squares).  Cross-row reduction: the 4 rows of a wave by lane exchange, the 16 waves through LDS, summed by every
// thread in the same fixed order.
constexpr int BNF_RPT = 16;
__device__ __forceinline__ float bnf_reduce(float v, float (*sm)[16], int tx, int wv) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  __syncthreads();                                  // the previous use of sm is over
  if ((threadIdx.x & 63) < 16) sm[wv][tx] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += sm[i][tx];
  return t;
}

// PARTS: y arrives as four partial matrices `pstride` floats apart (the K quarters of the in-pass value projection,
// ep_inpass.h), summed here as (p0 + p1) + (p2 + p3); the sum is written to y_out for the second token pass.
// I0 (PARTS only): rows [0, 64 I0) are already summed and sit in y_out (the whole-y tasks of the in-pass projection),
// rows behind them arrive as the four partials.
template <bool PARTS, int I0 = 0>
__global__ __launch_bounds__(1024) void ep_bn_fused_kernel(const float* __restrict__ y, int B, int Dp, float eps, float momentum,
                                                         float* __restrict__ z, float* __restrict__ rstd_out,
                                                         float* __restrict__ rmean, float* __restrict__ rvar,
                                                         int64_t* __restrict__ nbt, int64_t pstride,
                                                         float* __restrict__ y_out) {
  __shared__ float sm[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 16 + tx;
  const bool ok = col < Dp;
  float v[BNF_RPT];
  float s = 0.f;
  if constexpr (PARTS) {
    float pq[BNF_RPT - I0][4];
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {                // branch-free loads (clamped address): all in flight together
      const int b = ty + 64 * i;
      const int64_t at = (int64_t)(b < B ? b : B - 1) * Dp + (ok ? col : 0);
      if (i < I0) v[i] = y_out[at];
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) pq[i - I0][k] = y[(int64_t)k * pstride + at];
      }
    }
#pragma unroll
    for (int i = I0; i < BNF_RPT; ++i) v[i] = (pq[i - I0][0] + pq[i - I0][1]) + (pq[i - I0][2] + pq[i - I0][3]);
    if (y_out) {
#pragma unroll
      for (int i = I0; i < BNF_RPT; ++i) {
        const int b = ty + 64 * i;
        if (ok && b < B) y_out[(int64_t)b * Dp + col] = v[i];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {                  // branch-free loads (clamped address): all in flight together
      const int b = ty + 64 * i;
      v[i] = y[(int64_t)(b < B ? b : B - 1) * Dp + (ok ? col : 0)];
    }
  }
#pragma unroll
  for (int i = 0; i < BNF_RPT; ++i) {
    v[i] = (ok && ty + 64 * i < B) ? v[i] : 0.f;
    s += v[i];
  }
  const float mu = bnf_reduce(s, sm, tx, wv) / (float)B;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < BNF_RPT; ++i) {
    const float d = (ty + 64 * i < B) ? v[i] - mu : 0.f;
    q = fmaf(d, d, q);
  }
  const float var = bnf_reduce(q, sm, tx, wv) / (float)B;                 // biased: used for normalisation
  const float rs = 1.0f / sqrtf(var + eps);
  if (ty == 0 && ok) {
    rstd_out[col] = rs;
    const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
    rmean[col] = (1.0f - momentum) * rmean[col] + momentum * mu;
    rvar[col] = (1.0f - momentum) * rvar[col] + momentum * unbiased;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
#pragma unroll
  for (int i = 0; i < BNF_RPT; ++i) {
    const int b = ty + 64 * i;
    if (ok && b < B) z[(int64_t)b * Dp + col] = (v[i] - mu) * rs;
  }
}

__global__ __launch_bounds__(1024) void ep_bn_bwd_fused_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                             const float* __restrict__ rstd, int B, int Dp,
                                                             float* __restrict__ dy) {
  __shared__ float sm[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 16 + tx;
  const bool ok = col < Dp;
  float g[BNF_RPT], zz[BNF_RPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < BNF_RPT; ++i) {
    const int b = ty + 64 * i;
    const int64_t at = (int64_t)(b < B ? b : B - 1) * Dp + (ok ? col : 0);   // branch-free loads, all in flight together
    g[i] = dz[at]; zz[i] = z[at];
  }
#pragma unroll
  for (int i = 0; i < BNF_RPT; ++i) {
    const bool in = ok && ty + 64 * i < B;
    g[i] = in ? g[i] : 0.f;
    zz[i] = in ? zz[i] : 0.f;
    s1 += g[i];
    s2 = fmaf(g[i], zz[i], s2);
  }
  float rs = rstd[ok ? col : 0];
  const float m1 = bnf_reduce(s1, sm, tx, wv) / (float)B;
  const float m2 = bnf_reduce(s2, sm, tx, wv) / (float)B;
  // rs is pinned in a register HERE: a load result first used inside the guarded blocks below makes the compiler wait
  // vmcnt(0) in each of them -- i.e. for the round trip of the previous store (vmcnt counts loads and stores in order)
  asm volatile("" : "+v"(rs));
#pragma unroll
  for (int i = 0; i < BNF_RPT; ++i) {
    const int b = ty + 64 * i;
    if (ok && b < B) dy[(int64_t)b * Dp + col] = rs * (g[i] - m1 - zz[i] * m2);
  }
}

// Many rows (B > 4096: the BatchNorm over all B N token rows of the DOLG head): the same two-stage scheme with 1024-thread
// workgroups of 64 column lanes x 16 row lanes -- 256-byte row segments instead of 64-byte ones.
constexpr int BNW_C = 64, BNW_R = 16;
__device__ __forceinline__ float bnw_reduce(float v, float (*sm)[BNW_C], int tx, int ty) {
  __syncthreads();
  sm[ty][tx] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < BNW_R; ++i) t += sm[i][tx];
  return t;
}
__global__ __launch_bounds__(1024) void ep_bnw_stats_kernel(const float* __restrict__ y, int B, int Dp, float* __restrict__ partial) {
  __shared__ float sm[BNW_R][BNW_C];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * BNW_C + tx;
  const bool ok = col < Dp;
  int r0, r1;
  chunk_rows(B, gridDim.y, blockIdx.y, r0, r1);
  const int n = r1 - r0;
  float s = 0.f;
  if (ok) for (int b = r0 + ty; b < r1; b += BNW_R) s += y[(int64_t)b * Dp + col];
  const float mu = n > 0 ? bnw_reduce(s, sm, tx, ty) / (float)n : 0.f;
  float q = 0.f;
  if (ok) for (int b = r0 + ty; b < r1; b += BNW_R) { const float d = y[(int64_t)b * Dp + col] - mu; q = fmaf(d, d, q); }
  q = bnw_reduce(q, sm, tx, ty);
  if (ty == 0 && ok) {
    partial[((int64_t)blockIdx.y * 2 + 0) * Dp + col] = mu;
    partial[((int64_t)blockIdx.y * 2 + 1) * Dp + col] = q;
  }
}
__global__ __launch_bounds__(1024) void ep_bnw_apply_kernel(const float* __restrict__ y, int B, int Dp, float eps, float momentum,
                                                          const float* __restrict__ partial, float* __restrict__ z,
                                                          float* __restrict__ rstd_out, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, int64_t* __restrict__ nbt) {
  __shared__ float bc[2][BNW_C];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * BNW_C + tx;
  const bool ok = col < Dp;
  const int nrs = gridDim.y;
  if (ty == 0 && ok) {
    float tot = 0.f;
    for (int r = 0; r < nrs; ++r) {
      int a0, a1; chunk_rows(B, nrs, r, a0, a1);
      tot += (float)(a1 - a0) * partial[((int64_t)r * 2 + 0) * Dp + col];
    }
    const float mu = tot / (float)B;
    float m2 = 0.f;
    for (int r = 0; r < nrs; ++r) {
      int a0, a1; chunk_rows(B, nrs, r, a0, a1);
      const float d = partial[((int64_t)r * 2 + 0) * Dp + col] - mu;
      m2 += partial[((int64_t)r * 2 + 1) * Dp + col] + (float)(a1 - a0) * d * d;
    }
    const float var = m2 / (float)B;
    const float rs = 1.0f / sqrtf(var + eps);
    bc[0][tx] = mu; bc[1][tx] = rs;
    if (blockIdx.y == 0) {
      rstd_out[col] = rs;
      const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
      rmean[col] = (1.0f - momentum) * rmean[col] + momentum * mu;
      rvar[col] = (1.0f - momentum) * rvar[col] + momentum * unbiased;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  __syncthreads();
  const float mu = bc[0][tx], rs = bc[1][tx];
  int r0, r1;
  chunk_rows(B, nrs, blockIdx.y, r0, r1);
  if (ok) for (int b = r0 + ty; b < r1; b += BNW_R) z[(int64_t)b * Dp + col] = (y[(int64_t)b * Dp + col] - mu) * rs;
}
__global__ __launch_bounds__(1024) void ep_bnw_bwd_stats_kernel(const float* __restrict__ dz, const float* __restrict__ z, int B, int Dp,
                                                              float* __restrict__ partial) {
  __shared__ float sm[BNW_R][BNW_C];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * BNW_C + tx;
  const bool ok = col < Dp;
  int r0, r1;
  chunk_rows(B, gridDim.y, blockIdx.y, r0, r1);
  float s1 = 0.f, s2 = 0.f;
  if (ok) for (int b = r0 + ty; b < r1; b += BNW_R) {
    const float g = dz[(int64_t)b * Dp + col];
    s1 += g;
    s2 = fmaf(g, z[(int64_t)b * Dp + col], s2);
  }
  s1 = bnw_reduce(s1, sm, tx, ty);
  s2 = bnw_reduce(s2, sm, tx, ty);
  if (ty == 0 && ok) {
    partial[((int64_t)blockIdx.y * 2 + 0) * Dp + col] = s1;
    partial[((int64_t)blockIdx.y * 2 + 1) * Dp + col] = s2;
  }
}
// (dy may alias dz: every element is read before it is written by the same thread)
__global__ __launch_bounds__(1024) void ep_bnw_bwd_apply_kernel(const float* dz, const float* __restrict__ z,
                                                              const float* __restrict__ rstd, int B, int Dp,
                                                              const float* __restrict__ partial, float* dy) {
  __shared__ float bc[2][BNW_C];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * BNW_C + tx;
  const bool ok = col < Dp;
  const int nrs = gridDim.y;
  if (ty == 0 && ok) {
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < nrs; ++r) {
      s1 += partial[((int64_t)r * 2 + 0) * Dp + col];
      s2 += partial[((int64_t)r * 2 + 1) * Dp + col];
    }
    bc[0][tx] = s1 / (float)B; bc[1][tx] = s2 / (float)B;
  }
  __syncthreads();
  const float m1 = bc[0][tx], m2 = bc[1][tx];
  int r0, r1;
  chunk_rows(B, nrs, blockIdx.y, r0, r1);
  if (ok) {
    const float rs = rstd[col];
    for (int b = r0 + ty; b < r1; b += BNW_R) {
      const int64_t i = (int64_t)b * Dp + col;
      dy[i] = rs * (dz[i] - m1 - z[i] * m2);
    }
  }
}

__global__ void ep_bn_eval_kernel(const float* __restrict__ y, int64_t total, int Dp, float eps,
                                  const float* __restrict__ rmean, const float* __restrict__ rvar,
                                  float* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % Dp);
  z[i] = (y[i] - rmean[col]) / sqrtf(rvar[col] + eps);
}

__global__ __launch_bounds__(256) void ep_bn_bwd_stats_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                            int B, int Dp, float* __restrict__ partial) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  int r0, r1;
  chunk_rows(B, gridDim.y, blockIdx.y, r0, r1);
  float s1 = 0.f, s2 = 0.f;
  if (ok) for (int b = r0 + ty; b < r1; b += RL) {
    const float g = dz[(int64_t)b * Dp + col];
    s1 += g;
    s2 = fmaf(g, z[(int64_t)b * Dp + col], s2);
  }
  s1 = colreduce(s1, sm, tx, ty);
  s2 = colreduce(s2, sm, tx, ty);
  if (ty == 0 && ok) {
    partial[((int64_t)blockIdx.y * 2 + 0) * Dp + col] = s1;
    partial[((int64_t)blockIdx.y * 2 + 1) * Dp + col] = s2;
  }
}

__global__ __launch_bounds__(256) void ep_bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                            const float* __restrict__ rstd, int B, int Dp,
                                                            const float* __restrict__ partial, float* __restrict__ dy) {
  __shared__ float bc[2][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  const int nrs = gridDim.y;
  if (ty == 0 && ok) {
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < nrs; ++r) {
      s1 += partial[((int64_t)r * 2 + 0) * Dp + col];
      s2 += partial[((int64_t)r * 2 + 1) * Dp + col];
    }
    bc[0][tx] = s1 / (float)B; bc[1][tx] = s2 / (float)B;
  }
  __syncthreads();
  const float m1 = bc[0][tx], m2 = bc[1][tx];
  int r0, r1;
  chunk_rows(B, nrs, blockIdx.y, r0, r1);
  if (ok) {
    const float rs = rstd[col];
    for (int b = r0 + ty; b < r1; b += RL) {
      const int64_t i = (int64_t)b * Dp + col;
      dy[i] = rs * (dz[i] - m1 - z[i] * m2);
    }
  }
}

// out[col] (+)= sum_b src[b*ld + col]
__global__ __launch_bounds__(256) void ep_colsum_kernel(const float* __restrict__ src, int B, int ncol, int ld,
                                                      int accumulate, float* __restrict__ out) {
  __shared__ float sm[RL][CG];
  colsum_block(src, B, ncol, ld, accumulate, out, blockIdx.x, sm);
}

// delta[b,q] = sum_c dy[b, q*Dq + c] * y[b, q*Dq + c]  ->  ML[b,q,2]   (one wave per (b,q))
__global__ __launch_bounds__(256) void ep_delta_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                     int rows, int Dq, float* __restrict__ ML,
                                                     const float* __restrict__ bias, int Q) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  if (bias) {
    const float* bq = bias + (int64_t)(r % Q) * Dq;
    for (int c = lane; c < Dq; c += 64) s = fmaf(dy[(int64_t)r * Dq + c], y[(int64_t)r * Dq + c] - bq[c], s);
  } else
  for (int c = lane; c < Dq; c += 64) s = fmaf(dy[(int64_t)r * Dq + c], y[(int64_t)r * Dq + c], s);
  s = wave_sum(s);
  if (lane == 0) ML[(int64_t)r * 4 + 2] = s;
}

// One wave per row of logits; the row lives in registers (C <= 64*CE_RPT on the fast path) so the
// logits are read once and every exponential is evaluated once.
constexpr int CE_RPT = 16;

template <bool FAST>
__global__ __launch_bounds__(256) void ep_ce_kernel(const float* __restrict__ logits, int ldl,
                                                  const int64_t* __restrict__ targets, int B, int C,
                                                  float grad_scale, float* __restrict__ loss_rows,
                                                  float* __restrict__ dlogits, float* __restrict__ rowstat,
                                                  const float* __restrict__ scale_dev) {
  if (scale_dev) grad_scale *= *scale_dev;            // device-resident loss scale (GradScaler.scale(loss))
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const float* row = logits + (int64_t)b * ldl;
  // a label outside [0, C) (an ignore_index, a wrong nb_classes) must not become an out-of-bounds read: the row is
  // flagged (rowstat[3], which stops the training loop like a non-finite loss), contributes no loss and no gradient
  const int64_t tgt64 = targets[b];
  const bool tgt_ok = tgt64 >= 0 && tgt64 < (int64_t)C;
  const int tgt = tgt_ok ? (int)tgt64 : 0;
  const float tv = row[tgt];
  float v[FAST ? CE_RPT : 1];
  float mx = -INFINITY;
  int bad = tgt_ok ? 0 : 1, rank = 0;
  if (FAST) {
#pragma unroll
    for (int r = 0; r < CE_RPT; ++r) {
      const int k = lane + 64 * r;
      v[r] = k < C ? row[k] : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < CE_RPT; ++r) {
      const int k = lane + 64 * r;
      if (k < C) {
        mx = fmaxf(mx, v[r]);
        bad |= !(fabsf(v[r]) <= 3.4028234664e38f);
        rank += (v[r] > tv) || (v[r] == tv && k < tgt);
      }
    }
  } else {
    for (int k = lane; k < C; k += 64) {
      const float x = row[k];
      mx = fmaxf(mx, x);
      bad |= !(fabsf(x) <= 3.4028234664e38f);
      rank += (x > tv) || (x == tv && k < tgt);       // position of the target in a descending sort
    }
  }
  mx = wave_max(mx);
  float se = 0.f;
  if (FAST) {
#pragma unroll
    for (int r = 0; r < CE_RPT; ++r) { v[r] = expf(v[r] - mx); se += v[r]; }   // exp(-inf) = 0 for the pad
  } else {
    for (int k = lane; k < C; k += 64) se += expf(row[k] - mx);
  }
  se = wave_sum(se);
  const float fr = wave_sum((float)rank);
  const float fb = wave_sum((float)bad);
  const float lse = mx + logf(se);
  const float loss = tgt_ok ? lse - tv : 0.f;
  if (dlogits) {
    const float g = tgt_ok ? grad_scale / (float)B : 0.f, inv = 1.0f / se;
    float* drow = dlogits + (int64_t)b * ldl;
    if (FAST) {
#pragma unroll
      for (int r = 0; r < CE_RPT; ++r) {
        const int k = lane + 64 * r;
        if (k < ldl) drow[k] = k < C ? (v[r] * inv - (k == tgt ? 1.0f : 0.0f)) * g : 0.f;
      }
    } else {
      for (int k = lane; k < ldl; k += 64) {
        float d = 0.f;
        if (k < C) d = (expf(row[k] - mx) * inv - (k == tgt ? 1.0f : 0.0f)) * g;
        drow[k] = d;
      }
    }
  }
  if (lane == 0) {
    if (loss_rows) loss_rows[b] = loss;
    if (rowstat) {
      const float isbad = (fb > 0.5f || !(fabsf(loss) <= 3.4028234664e38f)) ? 1.f : 0.f;
      *reinterpret_cast<f4*>(rowstat + (int64_t)b * 4) =
          f4{loss / (float)B, fr < 0.5f ? 1.f : 0.f, fr < 4.5f ? 1.f : 0.f, isbad};
    }
  }
}

// stats[0..3] += sum_b rowstat[b][0..3]   (one workgroup, fixed order: reproducible)
__global__ __launch_bounds__(256) void ep_ce_stats_kernel(const float* __restrict__ rowstat, int B,
                                                        float* __restrict__ stats) {
  __shared__ f4 sm[4];
  ce_stats_block(rowstat, B, stats, sm);
}

// ------------------------------------------------------------------------------------------
static int bn_row_splits(int B) {
  int rs = B / 64;                       // ~64 rows per workgroup
  if (rs < 1) rs = 1;
  if (rs > RS_MAX) rs = RS_MAX;
  return rs;
}
// one-launch kernels for batches whose rows fit the registers of one workgroup (EP_BN_FUSED=0: the two-launch path)
static bool bn_fused(int B) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_BN_FUSED"); on = e ? atoi(e) : 1; }
  return on && B <= 64 * BNF_RPT;
}
size_t bn_workspace_bytes(int B, int Dp) { (void)B; return round_up((size_t)RS_MAX * 2 * Dp * sizeof(float), 256); }

bool bn_takes_parts(int B) { return bn_fused(B); }
// the one split of "rows already summed | rows as partials" the kernel is instantiated for: 768 | rest (B = 1024 images on
// 768 pooling workgroups)
int bn_parts_r0(int B, int r0) { return (bn_fused(B) && r0 == 768 && B > 768) ? 768 : 0; }

int bn_forward_train(const float* y, int B, int Dp, float eps, float momentum, float* z, float* rstd,
                     float* rmean, float* rvar, int64_t* nbt, float* partial, hipStream_t st, int nparts, int64_t pstride,
                     float* y_out, int r0) {
  EP_REQUIRE(nparts == 1 || (nparts == 4 && bn_fused(B)), EP_E_UNSUPPORTED, "BatchNorm over partial sums: 4 parts, B <= %d", 64 * BNF_RPT);
  EP_REQUIRE(r0 == 0 || (nparts == 4 && r0 == 768 && y_out), EP_E_UNSUPPORTED, "BatchNorm over partial sums: row split %d not instantiated", r0);
  if (bn_fused(B)) {
    if (nparts == 4 && r0 == 768)
      hipLaunchKernelGGL((ep_bn_fused_kernel<true, 12>), dim3((Dp + 15) / 16), dim3(1024), 0, st, y, B, Dp, eps, momentum, z, rstd, rmean,
                         rvar, nbt, pstride, y_out);
    else if (nparts == 4)   // (a 4-column-per-workgroup variant for the four-fold input measured 22 us against 14.5: 16-byte row segments)
      hipLaunchKernelGGL(ep_bn_fused_kernel<true>, dim3((Dp + 15) / 16), dim3(1024), 0, st, y, B, Dp, eps, momentum, z, rstd, rmean,
                         rvar, nbt, pstride, y_out);
    else
      hipLaunchKernelGGL(ep_bn_fused_kernel<false>, dim3((Dp + 15) / 16), dim3(1024), 0, st, y, B, Dp, eps, momentum, z, rstd, rmean,
                         rvar, nbt, (int64_t)0, nullptr);
    EP_LAUNCH_CHECK("ep_bn_fused_kernel");
    return 0;
  }
  if (B > 4096) {
    const dim3 gw((Dp + BNW_C - 1) / BNW_C, RS_MAX);
    hipLaunchKernelGGL(ep_bnw_stats_kernel, gw, dim3(1024), 0, st, y, B, Dp, partial);
    hipLaunchKernelGGL(ep_bnw_apply_kernel, gw, dim3(1024), 0, st, y, B, Dp, eps, momentum, partial, z, rstd, rmean, rvar, nbt);
    EP_LAUNCH_CHECK("ep_bnw_train kernels");
    return 0;
  }
  const dim3 grid((Dp + CG - 1) / CG, bn_row_splits(B));
  hipLaunchKernelGGL(ep_bn_stats_kernel, grid, dim3(256), 0, st, y, B, Dp, partial);
  hipLaunchKernelGGL(ep_bn_apply_kernel, grid, dim3(256), 0, st, y, B, Dp, eps, momentum, partial, z, rstd, rmean,
                     rvar, nbt);
  EP_LAUNCH_CHECK("ep_bn_train kernels");
  return 0;
}
int bn_forward_eval(const float* y, int B, int Dp, float eps, const float* rmean, const float* rvar, float* z,
                    hipStream_t st) {
  const int64_t total = (int64_t)B * Dp;
  hipLaunchKernelGGL(ep_bn_eval_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, total, Dp,
                     eps, rmean, rvar, z);
  EP_LAUNCH_CHECK("ep_bn_eval_kernel");
  return 0;
}
int bn_backward(const float* dz, const float* z, const float* rstd, int B, int Dp, float* dy, float* partial,
                hipStream_t st) {
  if (bn_fused(B)) {
    hipLaunchKernelGGL(ep_bn_bwd_fused_kernel, dim3((Dp + 15) / 16), dim3(1024), 0, st, dz, z, rstd, B, Dp, dy);
    EP_LAUNCH_CHECK("ep_bn_bwd_fused_kernel");
    return 0;
  }
  if (B > 4096) {
    const dim3 gw((Dp + BNW_C - 1) / BNW_C, RS_MAX);
    hipLaunchKernelGGL(ep_bnw_bwd_stats_kernel, gw, dim3(1024), 0, st, dz, z, B, Dp, partial);
    hipLaunchKernelGGL(ep_bnw_bwd_apply_kernel, gw, dim3(1024), 0, st, dz, z, rstd, B, Dp, partial, dy);
    EP_LAUNCH_CHECK("ep_bnw_bwd kernels");
    return 0;
  }
  const dim3 grid((Dp + CG - 1) / CG, bn_row_splits(B));
  hipLaunchKernelGGL(ep_bn_bwd_stats_kernel, grid, dim3(256), 0, st, dz, z, B, Dp, partial);
  hipLaunchKernelGGL(ep_bn_bwd_apply_kernel, grid, dim3(256), 0, st, dz, z, rstd, B, Dp, partial, dy);
  EP_LAUNCH_CHECK("ep_bn_bwd kernels");
  return 0;
}
int colsum(const float* src, int B, int ncol, int ld, int accumulate, float* out, hipStream_t st) {
  hipLaunchKernelGGL(ep_colsum_kernel, dim3((ncol + CG - 1) / CG), dim3(256), 0, st, src, B, ncol, ld,
                     accumulate, out);
  EP_LAUNCH_CHECK("ep_colsum_kernel");
  return 0;
}
int delta_rows(const float* dy, const float* y, int rows, int Dq, float* ML, hipStream_t st, const float* bias, int Q) {
  hipLaunchKernelGGL(ep_delta_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dy, y, rows, Dq, ML, bias, Q);
  EP_LAUNCH_CHECK("ep_delta_kernel");
  return 0;
}
// dP[b, q, :] = sum_{c < DQ} dy[b, q DQ + c] Wv[q DQ + c, :]  (autograd of the value projection, reference poolings/ep.py:40)
// for THIN query slices, DQ = D' / Q <= 32 -- the published protocol's 32 queries at D <= 1024.  As a batched contraction on
// the matrix kernels (K = 24 at 256 x 768) it ran 84 us for 1.2 GFLOP (round 5 timeline, profiles/r05); it is bound by the 100 MB
// it writes.  One workgroup per (query, block of images): a thread keeps the DQ x 4 weights of its four output channels in
// registers, the dy slice of an image is wave-uniform (scalar loads), every output row is one coalesced float4 store per lane.
template <int DQ>
__global__ __launch_bounds__(256) void ep_dp_thin_kernel(const float* __restrict__ dy, const float* __restrict__ Wv, int B, int D, int Dp,
                                                        int Q, int rows_per_block, float* __restrict__ dP) {
  const int q = blockIdx.x, b0 = blockIdx.y * rows_per_block;
  const int c4 = threadIdx.x;
  if (4 * c4 >= D) return;
  f4 w[DQ];
#pragma unroll
  for (int c = 0; c < DQ; ++c) w[c] = *reinterpret_cast<const f4*>(Wv + (int64_t)(q * DQ + c) * D + 4 * c4);
  const int b1 = (b0 + rows_per_block) < B ? (b0 + rows_per_block) : B;
  for (int b = b0; b < b1; ++b) {
    const float* dyr = dy + (int64_t)b * Dp + q * DQ;
    f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < DQ; ++c) a += dyr[c] * w[c];
    *reinterpret_cast<f4*>(dP + ((int64_t)b * Q + q) * D + 4 * c4) = a;
  }
}
bool project_dp_thin_ok(int D, int Dp, int Q) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_DP_THIN"); on = e ? atoi(e) : 1; }
  const int Dq = Dp / Q;
  return on && D % 4 == 0 && D <= 1024 && Dp % Q == 0 && (Dq == 8 || Dq == 12 || Dq == 16 || Dq == 24 || Dq == 32);
}
int project_dp_thin(const float* dy, const float* Wv, int B, int D, int Dp, int Q, float* dP, hipStream_t st) {
  const int Dq = Dp / Q;
  const int rpb = 32;
  const dim3 grid(Q, (B + rpb - 1) / rpb), block(256);
#define EP_DPT(N_) case N_: hipLaunchKernelGGL(ep_dp_thin_kernel<N_>, grid, block, 0, st, dy, Wv, B, D, Dp, Q, rpb, dP); break;
  switch (Dq) {
    EP_DPT(8) EP_DPT(12) EP_DPT(16) EP_DPT(24) EP_DPT(32)
    default: set_error("project_dp_thin: Dq = %d", Dq); return EP_E_UNSUPPORTED;
  }
#undef EP_DPT
  EP_LAUNCH_CHECK("ep_dp_thin_kernel");
  return 0;
}
int cross_entropy(const float* logits, int ldl, const int64_t* targets, int B, int C, float grad_scale,
                  float* loss_rows, float* dlogits, float* rowstat, hipStream_t st, const float* scale_dev) {
  if (ldl <= 64 * CE_RPT)
    hipLaunchKernelGGL(ep_ce_kernel<true>, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, targets, B, C,
                       grad_scale, loss_rows, dlogits, rowstat, scale_dev);
  else
    hipLaunchKernelGGL(ep_ce_kernel<false>, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, targets, B, C,
                       grad_scale, loss_rows, dlogits, rowstat, scale_dev);
  EP_LAUNCH_CHECK("ep_ce_kernel");
  return 0;
}
int ce_stats(const float* rowstat, int B, float* stats, hipStream_t st) {
  hipLaunchKernelGGL(ep_ce_stats_kernel, dim3(1), dim3(256), 0, st, rowstat, B, stats);
  EP_LAUNCH_CHECK("ep_ce_stats_kernel");
  return 0;
}

}  // namespace ep
