// The small dense tail of the EP head: BatchNorm1d(affine=False) train/eval/backward
// (reference probe_heads.py:109-110), CrossEntropyLoss + top-k accuracy
// (reference main_linprobe.py:589, engine_finetune.py:62-63), bias gradient, and the softmax
// correction term delta of the pooling backward.  All fp32; batch reductions are done in a
// fixed order (no atomics on values that feed the parameters), so a step is reproducible.
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

constexpr int CG = 16;   // columns per workgroup
constexpr int RL = 16;   // row lanes per workgroup  (CG*RL = 256 threads)

// sum over the RL row-lanes of a workgroup for each of its CG columns; result valid for ty == 0
__device__ __forceinline__ float colreduce(float v, float (*sm)[CG], int tx, int ty) {
  sm[ty][tx] = v;
  __syncthreads();
  float s = 0.f;
  if (ty == 0) {
#pragma unroll
    for (int r = 0; r < RL; ++r) s += sm[r][tx];
  }
  __syncthreads();
  return s;
}

// Register-resident column tile: a workgroup owns CG columns and ALL B rows (B <= RL*RPT); each
// thread keeps its RPT rows of one column in registers, so y is read from memory exactly once and
// all loads of a thread are in flight together.
constexpr int RPT = 64;   // rows per thread  -> B <= 1024 on the fast path

template <bool FAST>
__global__ __launch_bounds__(256) void ep_bn_train_kernel(const float* __restrict__ y, int B, int Dp,
                                                        float eps, float momentum,
                                                        float* __restrict__ z, float* __restrict__ rstd_out,
                                                        float* __restrict__ rmean, float* __restrict__ rvar,
                                                        int64_t* __restrict__ nbt) {
  __shared__ float sm[RL][CG];
  __shared__ float bc[2][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  float v[FAST ? RPT : 1];
  float s = 0.f;
  if (FAST) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int b = ty + r * RL;
      v[r] = (ok && b < B) ? y[(int64_t)b * Dp + col] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) s += v[r];
  } else {
    if (ok) for (int b = ty; b < B; b += RL) s += y[(int64_t)b * Dp + col];
  }
  s = colreduce(s, sm, tx, ty);
  if (ty == 0) bc[0][tx] = s / (float)B;
  __syncthreads();
  const float mu = bc[0][tx];
  float q = 0.f;
  if (FAST) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) { const float d = (ty + r * RL < B) ? v[r] - mu : 0.f; q = fmaf(d, d, q); }
  } else {
    if (ok) for (int b = ty; b < B; b += RL) { const float d = y[(int64_t)b * Dp + col] - mu; q = fmaf(d, d, q); }
  }
  q = colreduce(q, sm, tx, ty);
  if (ty == 0) {
    const float var = q / (float)B;                      // biased: used for normalisation
    const float rs = 1.0f / sqrtf(var + eps);
    bc[1][tx] = rs;
    if (ok) {
      rstd_out[col] = rs;
      const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
      rmean[col] = (1.0f - momentum) * rmean[col] + momentum * mu;
      rvar[col] = (1.0f - momentum) * rvar[col] + momentum * unbiased;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  __syncthreads();
  const float rs = bc[1][tx];
  if (FAST) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int b = ty + r * RL;
      if (ok && b < B) z[(int64_t)b * Dp + col] = (v[r] - mu) * rs;
    }
  } else {
    if (ok) for (int b = ty; b < B; b += RL) z[(int64_t)b * Dp + col] = (y[(int64_t)b * Dp + col] - mu) * rs;
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

template <bool FAST>
__global__ __launch_bounds__(256) void ep_bn_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                      const float* __restrict__ rstd, int B, int Dp,
                                                      float* __restrict__ dy) {
  __shared__ float sm[RL][CG];
  __shared__ float bc[2][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  float g[FAST ? RPT : 1], zz[FAST ? RPT : 1];
  float s1 = 0.f, s2 = 0.f;
  if (FAST) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int b = ty + r * RL;
      const bool in = ok && b < B;
      g[r] = in ? dz[(int64_t)b * Dp + col] : 0.f;
      zz[r] = in ? z[(int64_t)b * Dp + col] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) { s1 += g[r]; s2 = fmaf(g[r], zz[r], s2); }
  } else if (ok) {
    for (int b = ty; b < B; b += RL) {
      const float gv = dz[(int64_t)b * Dp + col];
      s1 += gv;
      s2 = fmaf(gv, z[(int64_t)b * Dp + col], s2);
    }
  }
  s1 = colreduce(s1, sm, tx, ty);
  s2 = colreduce(s2, sm, tx, ty);
  if (ty == 0) { bc[0][tx] = s1 / (float)B; bc[1][tx] = s2 / (float)B; }
  __syncthreads();
  const float m1 = bc[0][tx], m2 = bc[1][tx];
  if (ok) {
    const float rs = rstd[col];
    if (FAST) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int b = ty + r * RL;
        if (b < B) dy[(int64_t)b * Dp + col] = rs * (g[r] - m1 - zz[r] * m2);
      }
    } else {
      for (int b = ty; b < B; b += RL) {
        const int64_t i = (int64_t)b * Dp + col;
        dy[i] = rs * (dz[i] - m1 - z[i] * m2);
      }
    }
  }
}

// out[col] (+)= sum_b src[b*ld + col]
__global__ __launch_bounds__(256) void ep_colsum_kernel(const float* __restrict__ src, int B, int ncol, int ld,
                                                      int accumulate, float* __restrict__ out) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < ncol) {
    int b = ty;
    for (; b + 3 * RL < B; b += 4 * RL) {
      s0 += src[(int64_t)b * ld + col];
      s1 += src[(int64_t)(b + RL) * ld + col];
      s2 += src[(int64_t)(b + 2 * RL) * ld + col];
      s3 += src[(int64_t)(b + 3 * RL) * ld + col];
    }
    for (; b < B; b += RL) s0 += src[(int64_t)b * ld + col];
  }
  float s = colreduce((s0 + s1) + (s2 + s3), sm, tx, ty);
  if (ty == 0 && col < ncol) out[col] = accumulate ? out[col] + s : s;
}

// delta[b,q] = sum_c dy[b, q*Dq + c] * y[b, q*Dq + c]  ->  ML[b,q,2]   (one wave per (b,q))
__global__ __launch_bounds__(256) void ep_delta_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                     int rows, int Dq, float* __restrict__ ML) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
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
                                                  float* __restrict__ dlogits, float* __restrict__ stats) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const float* row = logits + (int64_t)b * ldl;
  const int tgt = (int)targets[b];
  const float tv = row[tgt];
  float v[FAST ? CE_RPT : 1];
  float mx = -INFINITY;
  int bad = 0, rank = 0;
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
  const float loss = lse - tv;
  if (dlogits) {
    const float g = grad_scale / (float)B, inv = 1.0f / se;
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
    if (stats) {
      atomicAdd(&stats[0], loss / (float)B);
      if (fr < 0.5f) atomicAdd(&stats[1], 1.0f);
      if (fr < 4.5f) atomicAdd(&stats[2], 1.0f);
      if (fb > 0.5f || !(fabsf(loss) <= 3.4028234664e38f)) atomicAdd(&stats[3], 1.0f);
    }
  }
}

// ------------------------------------------------------------------------------------------
int bn_forward_train(const float* y, int B, int Dp, float eps, float momentum, float* z, float* rstd,
                     float* rmean, float* rvar, int64_t* nbt, hipStream_t st) {
  if (B <= RL * RPT)
    hipLaunchKernelGGL(ep_bn_train_kernel<true>, dim3((Dp + CG - 1) / CG), dim3(256), 0, st, y, B, Dp, eps,
                       momentum, z, rstd, rmean, rvar, nbt);
  else
    hipLaunchKernelGGL(ep_bn_train_kernel<false>, dim3((Dp + CG - 1) / CG), dim3(256), 0, st, y, B, Dp, eps,
                       momentum, z, rstd, rmean, rvar, nbt);
  EP_LAUNCH_CHECK("ep_bn_train_kernel");
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
int bn_backward(const float* dz, const float* z, const float* rstd, int B, int Dp, float* dy, hipStream_t st) {
  if (B <= RL * RPT)
    hipLaunchKernelGGL(ep_bn_bwd_kernel<true>, dim3((Dp + CG - 1) / CG), dim3(256), 0, st, dz, z, rstd, B, Dp, dy);
  else
    hipLaunchKernelGGL(ep_bn_bwd_kernel<false>, dim3((Dp + CG - 1) / CG), dim3(256), 0, st, dz, z, rstd, B, Dp, dy);
  EP_LAUNCH_CHECK("ep_bn_bwd_kernel");
  return 0;
}
int colsum(const float* src, int B, int ncol, int ld, int accumulate, float* out, hipStream_t st) {
  hipLaunchKernelGGL(ep_colsum_kernel, dim3((ncol + CG - 1) / CG), dim3(256), 0, st, src, B, ncol, ld,
                     accumulate, out);
  EP_LAUNCH_CHECK("ep_colsum_kernel");
  return 0;
}
int delta_rows(const float* dy, const float* y, int rows, int Dq, float* ML, hipStream_t st) {
  hipLaunchKernelGGL(ep_delta_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dy, y, rows, Dq, ML);
  EP_LAUNCH_CHECK("ep_delta_kernel");
  return 0;
}
int cross_entropy(const float* logits, int ldl, const int64_t* targets, int B, int C, float grad_scale,
                  float* loss_rows, float* dlogits, float* stats, hipStream_t st) {
  if (ldl <= 64 * CE_RPT)
    hipLaunchKernelGGL(ep_ce_kernel<true>, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, targets, B, C,
                       grad_scale, loss_rows, dlogits, stats);
  else
    hipLaunchKernelGGL(ep_ce_kernel<false>, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, targets, B, C,
                       grad_scale, loss_rows, dlogits, stats);
  EP_LAUNCH_CHECK("ep_ce_kernel");
  return 0;
}

}  // namespace ep
