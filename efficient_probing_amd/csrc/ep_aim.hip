// AIM attention-pooling head (reference poolings/aim.py:337-392 AttentionPoolingClassifier, registry entry
// probe_heads.py:73: AttentionPoolingClassifier(dim=dim, num_heads=args.num_heads)) on the EP token passes.
//
// The reference batch-normalises every token per channel (BatchNorm1d(dim, affine=False, eps=1e-6) over the B*N
// tokens of the batch, aim.py:357,364), projects the normalised tokens to keys and values (k, v: Linear(dim, dim),
// no bias) and lets ONE learned query token, split into H heads and scaled, attend over them (aim.py:371-391).
// With the channel statistics mu, r = 1/sqrt(var + eps) (constants w.r.t. the parameters: the tokens are frozen
// and the normalisation has no affine part) and xt = (x - mu) * r:
//     score[b,h,n] = (scale q_h) . (Wk_h xt[b,n]) = (r * u_h) . x[b,n] + const,   u_h = scale Wk_h^T q_h
//     out[b,h]     = Wv_h (sum_n A[b,h,n] xt[b,n]) = (Wv diag(r))_h P[b,h] - Wv_h (mu * r),   P = sum_n A x
// so the token-dependent part is the plain EP pooling pass with the H derived query rows w_h = r * u_h (the constant
// cancels in the softmax), followed by EP's per-query projection with the folded weight Wv diag(r) and the bias
// -Wv (mu * r).  The channel statistics need one more streaming read of the batch -- or none: the per-image
// column statistics {mean, M2 over the N tokens} depend on the frozen tokens only, so a resident token store
// computes them once (ep_channel_stats) and a step combines the B cached rows (Chan's formula, fixed order).
#include "ep_side.h"
#include "ep_lnaffine.h"

namespace ep {

// Per-image sums over the tokens are taken as FOUR partial sums (tokens n = w, w + 4, ... in increasing n) combined as
// (p0 + p1) + (p2 + p3) -- the order shared by ep_chanstats_kernel, ep_xhat_mean_kernel (ep_clip.hip) and the fused
// ep_tokimg_kernel below, so cached tables and tables recomputed for a batch hold the same bits.
__device__ __forceinline__ void chan_finish(const f4& k, const f4& s1, const f4& s2, float inv, f4& mean, f4& m2) {
  mean = f4{__fadd_rn(k.x, __fmul_rn(s1.x, inv)), __fadd_rn(k.y, __fmul_rn(s1.y, inv)), __fadd_rn(k.z, __fmul_rn(s1.z, inv)),
            __fadd_rn(k.w, __fmul_rn(s1.w, inv))};
  m2 = f4{__fsub_rn(s2.x, __fmul_rn(__fmul_rn(s1.x, s1.x), inv)), __fsub_rn(s2.y, __fmul_rn(__fmul_rn(s1.y, s1.y), inv)),
          __fsub_rn(s2.z, __fmul_rn(__fmul_rn(s1.z, s1.z), inv)), __fsub_rn(s2.w, __fmul_rn(__fmul_rn(s1.w, s1.w), inv))};
}
__device__ __forceinline__ void chan_acc(f4& s1, f4& s2, const f4& v, const f4& k) {
  const f4 d = {__fsub_rn(v.x, k.x), __fsub_rn(v.y, k.y), __fsub_rn(v.z, k.z), __fsub_rn(v.w, k.w)};
  s1.x = __fadd_rn(s1.x, d.x); s1.y = __fadd_rn(s1.y, d.y); s1.z = __fadd_rn(s1.z, d.z); s1.w = __fadd_rn(s1.w, d.w);
  s2.x = fmaf(d.x, d.x, s2.x); s2.y = fmaf(d.y, d.y, s2.y); s2.z = fmaf(d.z, d.z, s2.z); s2.w = fmaf(d.w, d.w, s2.w);
}
__device__ __forceinline__ f4 sum4parts(const f4& a, const f4& b, const f4& c, const f4& d) {
  return f4{__fadd_rn(__fadd_rn(a.x, b.x), __fadd_rn(c.x, d.x)), __fadd_rn(__fadd_rn(a.y, b.y), __fadd_rn(c.y, d.y)),
            __fadd_rn(__fadd_rn(a.z, b.z), __fadd_rn(c.z, d.z)), __fadd_rn(__fadd_rn(a.w, b.w), __fadd_rn(c.w, d.w))};
}

// per image and channel, shifted by the first token k: mean = k + s1 / N, M2 = s2 - s1^2 / N with s1 = sum (x - k),
// s2 = sum (x - k)^2     (a thread owns 4 channels and walks the tokens)
template <bool BF16>
__global__ __launch_bounds__(256) void ep_chanstats_kernel(const void* __restrict__ x, int64_t bstride,
                                                         const int* __restrict__ index, int N, int D,
                                                         float* __restrict__ img) {
  const int b = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= D) return;
  const int64_t e0 = (int64_t)(index ? index[b] : b) * bstride + c;
  const f4 k = load_tok4<BF16>(x, e0);
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  f4 s1[4] = {z, z, z, z}, s2[4] = {z, z, z, z};
  int n0 = 0;
  for (; n0 + 8 <= N; n0 += 8) {                       // eight independent row loads in flight
    f4 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = load_tok4<BF16>(x, e0 + (int64_t)(n0 + t) * D);
#pragma unroll
    for (int t = 0; t < 8; ++t) chan_acc(s1[t & 3], s2[t & 3], v[t], k);
  }
  for (; n0 < N; n0 += 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (n0 + t < N) chan_acc(s1[t], s2[t], load_tok4<BF16>(x, e0 + (int64_t)(n0 + t) * D), k);
  }
  const f4 a1 = sum4parts(s1[0], s1[1], s1[2], s1[3]), a2 = sum4parts(s2[0], s2[1], s2[2], s2[3]);
  const float inv = 1.0f / (float)N;
  f4 mean, m2;
  chan_finish(k, a1, a2, inv, mean, m2);
  float* o = img + (int64_t)b * 2 * D + c;
  *reinterpret_cast<f4*>(o) = mean;
  *reinterpret_cast<f4*>(o + D) = m2;
}

// xhat accumulation of the fused kernel and of ep_xhat_mean_kernel: s += (x - mean) * rstd, one fma per component
__device__ __forceinline__ void xhat_acc(f4& s, const f4& v, float mean, float rstd) {
  s.x = fmaf(__fsub_rn(v.x, mean), rstd, s.x); s.y = fmaf(__fsub_rn(v.y, mean), rstd, s.y);
  s.z = fmaf(__fsub_rn(v.z, mean), rstd, s.z); s.w = fmaf(__fsub_rn(v.w, mean), rstd, s.w);
}

// ONE read of the tokens for the per-token LayerNorm statistics AND a per-image table: one 4-wave workgroup per image, wave w
// takes the tokens n = w, w + 4, ... with the row in registers (CPL 16-byte chunks per lane) -- the statistics exactly as
// ep_token_stats_reg_kernel computes them -- and every lane accumulates its channels:
//   MODE 0: the shifted channel sums of ep_chanstats_kernel -> img (B, 2, D)        (SimPool: the mean token)
//   MODE 1: sum_n xhat -> xbar (B, D) = mean_n xhat                                  (CLIP: the mean row)
template <bool BF16, int CPL, int MODE>
__global__ __launch_bounds__(256) void ep_tokimg_kernel(const void* __restrict__ x, int64_t bstride, int N, int D, float eps,
                                                      float* __restrict__ stats, float* __restrict__ out) {
  constexpr int NA = MODE == 0 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float part[4 * 64 * CPL * NA * 4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e0 = (int64_t)b * bstride;
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  f4 k[CPL], a1[CPL], a2[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int d = lane * 4 + 256 * j;
    k[j] = (MODE == 0 && d < D) ? load_tok4<BF16>(x, e0 + d) : z;
    a1[j] = z; a2[j] = z;
  }
  for (int n = w; n < N; n += 4) {
    f4 v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int d = lane * 4 + 256 * j;
      v[j] = d < D ? load_tok4<BF16>(x, e0 + (int64_t)n * D + d) : z;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (lane * 4 + 256 * j < D) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (lane * 4 + 256 * j < D) {
        const f4 c = v[j] - mean;
        q = fmaf(c.x, c.x, q); q = fmaf(c.y, c.y, q); q = fmaf(c.z, c.z, q); q = fmaf(c.w, c.w, q);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { stats[((int64_t)b * N + n) * 2] = mean; stats[((int64_t)b * N + n) * 2 + 1] = rstd; }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      if (lane * 4 + 256 * j >= D) continue;
      if (MODE == 0) chan_acc(a1[j], a2[j], v[j], k[j]);
      else xhat_acc(a1[j], v[j], mean, rstd);
    }
  }
  f4* mine = reinterpret_cast<f4*>(part) + ((size_t)w * 64 + lane) * CPL * NA;
#pragma unroll
  for (int j = 0; j < CPL; ++j) { mine[j * NA] = a1[j]; if (MODE == 0) mine[j * NA + 1] = a2[j]; }
  __syncthreads();
  if (w != 0) return;
  const f4* p0 = reinterpret_cast<const f4*>(part) + (size_t)lane * CPL * NA;
  const size_t ws = (size_t)64 * CPL * NA;
  const float inv = 1.0f / (float)N;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int d = lane * 4 + 256 * j;
    if (d >= D) continue;
    const f4 t1 = sum4parts(p0[j * NA], p0[ws + j * NA], p0[2 * ws + j * NA], p0[3 * ws + j * NA]);
    if (MODE == 0) {
      const f4 t2 = sum4parts(p0[j * NA + 1], p0[ws + j * NA + 1], p0[2 * ws + j * NA + 1], p0[3 * ws + j * NA + 1]);
      f4 mean, m2;
      chan_finish(k[j], t1, t2, inv, mean, m2);
      float* o = out + (int64_t)b * 2 * D + d;
      *reinterpret_cast<f4*>(o) = mean;
      *reinterpret_cast<f4*>(o + D) = m2;
    } else {
      *reinterpret_cast<f4*>(out + (int64_t)b * D + d) =
          f4{__fmul_rn(t1.x, inv), __fmul_rn(t1.y, inv), __fmul_rn(t1.z, inv), __fmul_rn(t1.w, inv)};
    }
  }
}

// ---- combine the B per-image rows: mu, r, nb = -mu r; running statistics (momentum, unbiased variance) -------------
// equal counts N per image:  mean = (1/B) sum_b mean_b ;  M2 = sum_b M2_b + N sum_b (mean_b - mean)^2
// 16 columns x 16 row-lanes per workgroup; every partial sum has a fixed order (reproducible).
__global__ __launch_bounds__(1024) void ep_chancombine_kernel(const float* __restrict__ img, const int* __restrict__ index,
                                                            int B, int N, int D, float eps, float momentum,
                                                            float* __restrict__ mu_out, float* __restrict__ r_out,
                                                            float* __restrict__ nb_out, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, int64_t* __restrict__ nbt) {
  __shared__ float sm[32][33];                       // 32 column lanes x 32 row lanes over the B image rows (grid (D + 31) / 32)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  const bool ok = c < D;
  auto colsum32 = [&](float v) {
    __syncthreads();
    sm[ty][tx] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    return t;
  };
  float s = 0.f;
  if (ok) for (int b = ty; b < B; b += 32) s += img[(int64_t)(index ? index[b] : b) * 2 * D + c];
  const float mean = colsum32(s) / (float)B;
  float q = 0.f;
  if (ok) for (int b = ty; b < B; b += 32) {
    const float* row = img + (int64_t)(index ? index[b] : b) * 2 * D;
    const float dm = row[c] - mean;
    q += fmaf((float)N * dm, dm, row[D + c]);
  }
  q = colsum32(q);
  if (ty == 0 && ok) {
    const float n = (float)B * (float)N;
    const float var = q / n;                                  // biased: used for the normalisation
    const float r = 1.0f / sqrtf(var + eps);
    mu_out[c] = mean; r_out[c] = r; nb_out[c] = -mean * r;
    if (rmean) {
      const float unbiased = n > 1.f ? q / (n - 1.f) : var;
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * unbiased;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
}

// eval mode: the running statistics normalise (aim.py:364 in model.eval())
__global__ void ep_aim_evalstats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int D, float eps,
                                        float* __restrict__ mu, float* __restrict__ r, float* __restrict__ nb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  const float rr = 1.0f / sqrtf(rvar[c] + eps);
  mu[c] = rmean[c]; r[c] = rr; nb[c] = -rmean[c] * rr;
}

// u[h,d] = scale sum_c cls[h dh + c] Wk[h dh + c, d] ;  wq[h,d] = r[d] u[h,d]
__global__ __launch_bounds__(1024) void ep_aim_w_kernel(const float* __restrict__ cls, const float* __restrict__ Wk,
                                                      const float* __restrict__ r, int D, int dh, float scale,
                                                      float* __restrict__ wq) {
  __shared__ float sm[32][33];                       // grid (D / 32, H): 32 column lanes x 32 row lanes over the head's dh rows
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx, h = blockIdx.y;
  float acc = 0.f;
  if (d < D)
    for (int c = ty; c < dh; c += 32) acc = fmaf(cls[h * dh + c], Wk[(int64_t)(h * dh + c) * D + d], acc);
  sm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    wq[(int64_t)h * D + d] = t * scale * r[d];
  }
}

// du[h,d] = r[d] dw[h,d] ;  dcls[j] (+)= scale Wk[j,:] . du[h(j),:]      (one wave per row j)
__global__ __launch_bounds__(256) void ep_aim_dcls_kernel(const float* __restrict__ dw, const float* __restrict__ Wk,
                                                        const float* __restrict__ r, int D, int dh, float scale,
                                                        int accumulate, float* __restrict__ dcls) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D) return;
  const int h = j / dh, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(Wk[(int64_t)j * D + d], r[d] * dw[(int64_t)h * D + d], acc);
  acc = wave_sum(acc) * scale;
  if (lane == 0) dcls[j] = accumulate ? dcls[j] + acc : acc;
}

// dWk[j,d] (+)= scale cls[j] r[d] dw[h(j),d]
__global__ __launch_bounds__(256) void ep_aim_dwk_kernel(const float* __restrict__ dw, const float* __restrict__ cls,
                                                       const float* __restrict__ r, int D, int dh, float scale,
                                                       int accumulate, float* __restrict__ dWk) {
  const int d = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (d >= D) return;
  const float v = scale * cls[j] * r[d] * dw[(int64_t)(j / dh) * D + d];
  float* o = dWk + (int64_t)j * D + d;
  *o = accumulate ? *o + v : v;
}

// ---------------------------------------------------------------------------------------------
constexpr int AIM_NT = 5;                            // cls_token | k.weight | v.weight | fc.weight | fc.bias
struct AimWs {
  float *P, *S, *ML, *dP, *img, *mu, *r, *nb, *wq, *dw, *Wvs, *bo, *dWvs, *dbo, *scr;
  void* pool_ws; size_t pool_ws_bytes;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static int64_t aim_offsets(const ep_aim_dims& d, int64_t offs[AIM_NT]) {
  const int64_t D = d.D;
  const int64_t sizes[AIM_NT] = {D, D * D, D * D, (int64_t)d.C * D, d.C};
  int64_t off = 0;
  for (int i = 0; i < AIM_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static AimWs aim_carve(const ep_aim_dims& d, void* base, bool head) {
  AimWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D;
  w.P = take(B * d.H * D); w.S = take(B * d.H * d.N); w.ML = take(B * d.H * 4); w.dP = take(B * d.H * D);
  w.img = take(B * 2 * D);
  w.mu = take(D); w.r = take(D); w.nb = take(D); w.wq = take((size_t)d.H * D); w.dw = take((size_t)d.H * D);
  w.Wvs = take(D * D); w.bo = take(D); w.dWvs = take(D * D); w.dbo = take(D); w.scr = take(2 * D);
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.H);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[AIM_NT];
    w.opt_ws_bytes = optim_workspace_bytes(aim_offsets(d, offs), AIM_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int aim_check(const ep_aim_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0, EP_E_ARG, "aim dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0, EP_E_SHAPE, "aim: D %% H == 0 and D/H, D multiples of 4");
  EP_REQUIRE(d.H <= 32, EP_E_UNSUPPORTED, "aim: more than 32 heads");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "aim head: C must be positive");
  return 0;
}

static int aim_params_ok(const ep_aim_params* p, const char* what) {
  EP_REQUIRE(p && p->cls_token && p->k_w && p->v_w, EP_E_ARG, "%s: null tensor", what);
  EP_REQUIRE(aligned16(p->cls_token) && aligned16(p->k_w) && aligned16(p->v_w), EP_E_ALIGN, "%s: tensors must be 16-byte aligned", what);
  return 0;
}

int channel_stats(const void* x, int x_dtype, int64_t bstride, const int32_t* index, int B, int N, int D, float* img,
                  hipStream_t st) {
  const dim3 grid(B, (D / 4 + 255) / 256);
  if (x_dtype == EP_DTYPE_BF16)
    hipLaunchKernelGGL(ep_chanstats_kernel<true>, grid, dim3(256), 0, st, x, bstride, index, N, D, img);
  else
    hipLaunchKernelGGL(ep_chanstats_kernel<false>, grid, dim3(256), 0, st, x, bstride, index, N, D, img);
  EP_LAUNCH_CHECK("ep_chanstats_kernel");
  return 0;
}

// token statistics (B, N, 2) + a per-image table in one read of a dense batch (mode 0: channel statistics (B, 2, D);
// mode 1: the mean normalised row (B, D)) for rows of up to 1280 values (the callers check: otherwise the two kernels)
int token_image_stats(const void* x, int x_dtype, int64_t bstride, int B, int N, int D, float eps, int mode, float* stats, float* out,
                      hipStream_t st) {
  const int cpl = (D + 255) / 256;
  EP_REQUIRE(cpl <= 5 && D % 4 == 0 && (mode == 0 || mode == 1), EP_E_UNSUPPORTED, "token_image_stats: D=%d mode=%d", D, mode);
  const bool bf = x_dtype == EP_DTYPE_BF16;
#define EP_TI(C_, M_)                                                                                                          \
  if (cpl == C_ && mode == M_) {                                                                                               \
    if (bf) hipLaunchKernelGGL((ep_tokimg_kernel<true, C_, M_>), dim3(B), dim3(256), 0, st, x, bstride, N, D, eps, stats, out);  \
    else hipLaunchKernelGGL((ep_tokimg_kernel<false, C_, M_>), dim3(B), dim3(256), 0, st, x, bstride, N, D, eps, stats, out);    \
  }
  EP_TI(1, 0) EP_TI(2, 0) EP_TI(3, 0) EP_TI(4, 0) EP_TI(5, 0) EP_TI(1, 1) EP_TI(2, 1) EP_TI(3, 1) EP_TI(4, 1) EP_TI(5, 1)
#undef EP_TI
  EP_LAUNCH_CHECK("ep_tokimg_kernel");
  return 0;
}

static PoolParams aim_pool_params(const ep_aim_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                                  const AimWs& w) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);
  p.cls = w.wq; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index;
  return p;
}

static GemmParams ag(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

struct AimBn {                                       // the pooling's own token BatchNorm (aim.py:357)
  const float* img_stats;                            // optional: cached per-image column statistics (M, 2, D), indexed like x
  int training;
  float eps, momentum;
  float *running_mean, *running_var;
  int64_t* nbt;
};

static int aim_forward_core(const ep_aim_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                            const AimBn& bn, const ep_aim_params& pr, const AimWs& w, float* y, hipStream_t st) {
  const int D = d.D, dh = D / d.H;
  const float scale = (float)pow((double)dh, -0.5);                        // aim.py:349-350
  if (bn.training) {
    const float* img = bn.img_stats;
    const int32_t* iidx = index;                                            // cached rows are indexed like the tokens
    if (!img) {
      EP_TRY(channel_stats(x, x_dtype, bstride, index, d.B, d.N, D, w.img, st));
      img = w.img; iidx = nullptr;
    }
    hipLaunchKernelGGL(ep_chancombine_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, img, iidx, d.B, d.N, D, bn.eps,
                       bn.momentum, w.mu, w.r, w.nb, bn.running_mean, bn.running_var, bn.nbt);
  } else {
    EP_REQUIRE(bn.running_mean && bn.running_var, EP_E_ARG, "aim eval: running statistics missing");
    hipLaunchKernelGGL(ep_aim_evalstats_kernel, dim3((D + 255) / 256), dim3(256), 0, st, bn.running_mean, bn.running_var, D,
                       bn.eps, w.mu, w.r, w.nb);
  }
  hipLaunchKernelGGL(ep_aim_w_kernel, dim3((D + 31) / 32, d.H), dim3(1024), 0, st, pr.cls_token, pr.k_w, w.r, D, dh, scale, w.wq);
  hipLaunchKernelGGL(ep_cae_wv_kernel, dim3((D + 3) / 4), dim3(256), 0, st, pr.v_w, w.r, w.nb, D, w.Wvs, w.bo,
                     (const float*)nullptr);
  EP_LAUNCH_CHECK("ep_aim query kernels");
  EP_TRY(pool_forward(aim_pool_params(d, x, x_dtype, bstride, index, w), st));
  GemmParams g = ag(w.P, (int64_t)d.H * D, w.Wvs, D, y, D, d.B, dh, D);               // y_h = P_h (Wv r)_h^T - (Wv mu r)_h
  g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh; g.bias = w.bo; g.sBiasz = dh;
  return gemm(true, true, g, d.H, st);
}

// gradients of cls_token, k.weight, v.weight from dy (B, D); `y` is the forward's output (for the softmax correction)
static int aim_backward_core(const ep_aim_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const ep_aim_params& pr, const float* y, const float* dy, const ep_aim_params& gr, int acc,
                             const AimWs& w, SideTasks sd, hipStream_t st, hipStream_t aux) {
  const int D = d.D, dh = D / d.H, B = d.B;
  const float scale = (float)pow((double)dh, -0.5);
  EP_TRY(colsum(dy, B, D, D, 0, w.dbo, st));                                            // d(-Wv mu r)
  EP_TRY(delta_rows(dy, y, B * d.H, dh, w.ML, st, w.bo, d.H));                          // dP . P (bias taken out)
  {
    GemmParams g = ag(dy, D, w.Wvs, D, w.dP, (int64_t)d.H * D, B, D, dh);               // dP[b,h] = dy[b,h] (Wv r)_h
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D;
    EP_TRY(gemm(true, false, g, d.H, st));
  }
  GemmParams gWv = ag(dy, D, w.P, (int64_t)d.H * D, w.dWvs, D, dh, D, B);               // d(Wv r)_h = dy_h^T P_h
  gWv.sAz = dh; gWv.extA = dh; gWv.sBz = D; gWv.extB = D; gWv.sCz = (int64_t)dh * D; gWv.side = 1;
  EP_REQUIRE(gemm_side_ok(gWv, false, false), EP_E_ALIGN, "aim: unaligned gradient contraction");
  side_add_gemm(sd, gWv, d.H);
  PoolParams p = aim_pool_params(d, x, x_dtype, bstride, index, w);
  p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
  if (pool_backward_takes_side(p)) {
    EP_TRY(pool_backward(p, w.dw, 0, st, &sd));
  } else {                                           // (kernel families without side workgroups: the aux stream)
    AuxSide ax;
    EP_TRY(aux_side_begin(ax, st, aux));
    EP_TRY(aux_side_before_pass(ax, sd));
    EP_TRY(pool_backward(p, w.dw, 0, st));
    EP_TRY(aux_side_join(ax));
  }
  // dWv = d(Wv r) diag(r) + dbo nb^T   (the unused r / nb gradients of the shared kernel go to scratch)
  hipLaunchKernelGGL(ep_cae_dwv_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.dWvs, w.dbo, pr.v_w, w.r, w.nb, D, acc,
                     gr.v_w, w.scr, w.scr + D, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(ep_aim_dwk_kernel, dim3((D + 255) / 256, D), dim3(256), 0, st, w.dw, pr.cls_token, w.r, D, dh, scale, acc,
                     gr.k_w);
  hipLaunchKernelGGL(ep_aim_dcls_kernel, dim3((D + 3) / 4), dim3(256), 0, st, w.dw, pr.k_w, w.r, D, dh, scale, acc,
                     gr.cls_token);
  EP_LAUNCH_CHECK("ep_aim backward kernels");
  return 0;
}

static ep_aim_params aim_views(float* base, const int64_t o[AIM_NT]) {
  ep_aim_params p;
  p.cls_token = base + o[0]; p.k_w = base + o[1]; p.v_w = base + o[2];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

int ep_channel_stats(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                     float* image_stats, ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, 1));
  EP_REQUIRE(image_stats && aligned16(image_stats), EP_E_ARG, "ep_channel_stats: output null or not 16-byte aligned");
  return channel_stats(x, x_dtype, x_bstride, image_index, B, N, D, image_stats, (hipStream_t)stream);
}

size_t ep_aim_pool_workspace_bytes(const ep_aim_dims* dims) {
  if (!dims || aim_check(*dims, false) != 0) return 0;
  return aim_carve(*dims, nullptr, false).total;
}

int ep_aim_pool_forward(const ep_aim_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                        const float* image_stats, int training, float bn_eps, float bn_momentum, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, const ep_aim_params* params, float* y, void* ws,
                        size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_aim_pool_forward: null pointer");
  EP_TRY(aim_check(*dims, false));
  EP_TRY(aim_params_ok(params, "ep_aim_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_aim_pool_forward: y / ws must be 16-byte aligned");
  const AimWs w = aim_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_aim_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  const AimBn bn{image_stats, training, bn_eps, bn_momentum, running_mean, running_var, num_batches_tracked};
  return aim_forward_core(*dims, x, x_dtype, x_bstride, image_index, bn, *params, w, y, (hipStream_t)stream);
}

int ep_aim_pool_backward(const ep_aim_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                         const ep_aim_params* params, const float* y, const float* dy, const ep_aim_params* grads,
                         int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && dy && ws, EP_E_ARG, "ep_aim_pool_backward: null pointer");
  EP_TRY(aim_check(*dims, false));
  EP_TRY(aim_params_ok(params, "ep_aim_pool_backward(params)"));
  EP_TRY(aim_params_ok(grads, "ep_aim_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dy) && aligned16(y), EP_E_ALIGN, "ep_aim_pool_backward: y / dy / ws must be 16-byte aligned");
  const AimWs w = aim_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_aim_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return aim_backward_core(*dims, x, x_dtype, x_bstride, image_index, *params, y, dy, *grads, accumulate, w, SideTasks{},
                           (hipStream_t)stream, nullptr);
}

int ep_aim_attention(const ep_aim_dims* dims, const void* ws, float* A, ep_stream_t stream) {
  EP_REQUIRE(dims && ws && A, EP_E_ARG, "ep_aim_attention: null pointer");
  EP_TRY(aim_check(*dims, false));
  const AimWs w = aim_carve(*dims, const_cast<void*>(ws), false);
  return attention_from_scores(w.S, w.ML, dims->B * dims->H, dims->N, A, (hipStream_t)stream);
}

int64_t ep_aim_head_param_offsets(const ep_aim_dims* dims, int64_t offsets[5]) { return aim_offsets(*dims, offsets); }

size_t ep_aim_head_workspace_bytes(const ep_aim_dims* dims) {
  if (!dims || aim_check(*dims, true) != 0) return 0;
  return aim_carve(*dims, nullptr, true).total;
}

int ep_aim_head_train_step(const ep_aim_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_aim_head_train_step: null pointer");
  const ep_aim_dims& d = s->dims;
  EP_TRY(aim_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const AimWs w = aim_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_aim_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[AIM_NT];
  const int64_t total = aim_offsets(d, offs);
  const ep_aim_params pr = aim_views(s->params, offs), gr = aim_views(s->grads, offs);
  float* Wc = s->params + offs[3]; float* bc = s->params + offs[4];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats && s->tok_running_mean &&
               s->tok_running_var, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    const AimBn bn{s->image_stats, 1, s->tok_bn_eps, s->tok_bn_momentum, s->tok_running_mean, s->tok_running_var,
                   s->tok_num_batches_tracked};
    EP_TRY(aim_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, bn, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, nullptr, nullptr, 0, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    SideTasks sd{};
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, d.D, d.C, s->grads + offs[3], s->accumulate);
    EP_REQUIRE(gemm_side_ok(gWc, false, false), EP_E_ALIGN, "aim head: unaligned classifier gradient");
    side_add_gemm(sd, gWc, 1);
    sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[4]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
    sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
    sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
    sd.total += sd.n_colsum + sd.n_stats;
    EP_TRY(aim_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, pr, w.y, w.dy, gr, s->accumulate, w, sd, st,
                             (hipStream_t)s->aux_stream));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t D = d.D;
    const int64_t sizes[AIM_NT] = {D, D * D, D * D, (int64_t)d.C * D, d.C};
    // util/lars.py:22: trust ratio + weight decay for ndim > 1: cls_token is (1, 1, D)
    const int trust[AIM_NT] = {1, 1, 1, 1, 0};
    ep_segment segs[AIM_NT];
    for (int i = 0; i < AIM_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? AIM_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_aim_head_eval_forward(const ep_aim_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                             float tok_bn_eps, const float* tok_running_mean, const float* tok_running_var,
                             const float* params, const float* running_mean, const float* running_var, float bn_eps,
                             float* logits, int ldl, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && tok_running_mean && tok_running_var && logits && ws,
             EP_E_ARG, "ep_aim_head_eval_forward: null pointer");
  const ep_aim_dims& d = *dims;
  EP_TRY(aim_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const AimWs w = aim_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_aim_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[AIM_NT];
  aim_offsets(d, offs);
  const ep_aim_params pr = aim_views(const_cast<float*>(params), offs);
  const AimBn bn{nullptr, 0, tok_bn_eps, 0.f, const_cast<float*>(tok_running_mean), const_cast<float*>(tok_running_var), nullptr};
  EP_TRY(aim_forward_core(d, x, x_dtype, x_bstride, image_index, bn, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[3], params + offs[4], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
