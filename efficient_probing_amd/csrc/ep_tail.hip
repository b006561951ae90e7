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
  float s = 0.f;
  if (ok) for (int b = ty; b < B; b += RL) s += y[(int64_t)b * Dp + col];
  s = colreduce(s, sm, tx, ty);
  if (ty == 0) bc[0][tx] = s / (float)B;
  __syncthreads();
  const float mu = bc[0][tx];
  float v = 0.f;
  if (ok) for (int b = ty; b < B; b += RL) { const float d = y[(int64_t)b * Dp + col] - mu; v = fmaf(d, d, v); }
  v = colreduce(v, sm, tx, ty);
  if (ty == 0) {
    const float var = v / (float)B;                      // biased: used for normalisation
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
  if (ok) for (int b = ty; b < B; b += RL) z[(int64_t)b * Dp + col] = (y[(int64_t)b * Dp + col] - mu) * rs;
}

__global__ void ep_bn_eval_kernel(const float* __restrict__ y, int64_t total, int Dp, float eps,
                                  const float* __restrict__ rmean, const float* __restrict__ rvar,
                                  float* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % Dp);
  z[i] = (y[i] - rmean[col]) / sqrtf(rvar[col] + eps);
}

__global__ __launch_bounds__(256) void ep_bn_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                      const float* __restrict__ rstd, int B, int Dp,
                                                      float* __restrict__ dy) {
  __shared__ float sm[RL][CG];
  __shared__ float bc[2][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  const bool ok = col < Dp;
  float s1 = 0.f, s2 = 0.f;
  if (ok) for (int b = ty; b < B; b += RL) {
    const float g = dz[(int64_t)b * Dp + col];
    s1 += g;
    s2 = fmaf(g, z[(int64_t)b * Dp + col], s2);
  }
  s1 = colreduce(s1, sm, tx, ty);
  s2 = colreduce(s2, sm, tx, ty);
  if (ty == 0) { bc[0][tx] = s1 / (float)B; bc[1][tx] = s2 / (float)B; }
  __syncthreads();
  const float m1 = bc[0][tx], m2 = bc[1][tx];
  if (ok) {
    const float rs = rstd[col];
    for (int b = ty; b < B; b += RL) {
      const int64_t i = (int64_t)b * Dp + col;
      dy[i] = rs * (dz[i] - m1 - z[i] * m2);
    }
  }
}

// out[col] (+)= sum_b src[b*ld + col]
__global__ __launch_bounds__(256) void ep_colsum_kernel(const float* __restrict__ src, int B, int ncol, int ld,
                                                      int accumulate, float* __restrict__ out) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = blockIdx.x * CG + tx;
  float s = 0.f;
  if (col < ncol) for (int b = ty; b < B; b += RL) s += src[(int64_t)b * ld + col];
  s = colreduce(s, sm, tx, ty);
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

// One wave per row of logits.
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
  float mx = -INFINITY;
  int bad = 0, rank = 0;
  for (int k = lane; k < C; k += 64) {
    const float v = row[k];
    mx = fmaxf(mx, v);
    bad |= !(fabsf(v) <= 3.4028234664e38f);
    rank += (v > tv) || (v == tv && k < tgt);       // position of the target in a descending sort
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < C; k += 64) se += expf(row[k] - mx);
  se = wave_sum(se);
  const float fr = wave_sum((float)rank);
  const float fb = wave_sum((float)bad);
  const float lse = mx + logf(se);
  const float loss = lse - tv;
  if (dlogits) {
    const float g = grad_scale / (float)B, inv = 1.0f / se;
    float* drow = dlogits + (int64_t)b * ldl;
    for (int k = lane; k < ldl; k += 64) {
      float d = 0.f;
      if (k < C) d = (expf(row[k] - mx) * inv - (k == tgt ? 1.0f : 0.0f)) * g;
      drow[k] = d;
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
  hipLaunchKernelGGL(ep_bn_train_kernel, dim3((Dp + CG - 1) / CG), dim3(256), 0, st, y, B, Dp, eps, momentum,
                     z, rstd, rmean, rvar, nbt);
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
  hipLaunchKernelGGL(ep_bn_bwd_kernel, dim3((Dp + CG - 1) / CG), dim3(256), 0, st, dz, z, rstd, B, Dp, dy);
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
  hipLaunchKernelGGL(ep_ce_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, targets, B, C, grad_scale,
                     loss_rows, dlogits, stats);
  EP_LAUNCH_CHECK("ep_ce_kernel");
  return 0;
}

}  // namespace ep
