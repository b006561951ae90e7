// AbMILP head (reference poolings/abmilp.py:11-75 + models_vit.py:43-97 Attention, one head, no qkv
// bias) -- the matrix-core-bound member of the probe-head family.
//
// forward per image (N x D tokens x):
//   QKV = x Wqkv^T                                   (models_vit.py:74)
//   A   = softmax_j((q * D^-1/2) k^T)                (models_vit.py:86-89, temperature 1)
//   Xa  = (A v) Wp^T + bp                            (models_vit.py:92-94)
//   a   = softmax_n(w2 . tanh(W1 Xa + b1) + b2)      (abmilp.py:44-52,62-63; self-attention applied to "both")
//   out = sum_n a[n] Xa[n]                           (abmilp.py:65-66)
// Every contraction is one call of the exact-fp32 MFMA kernel (ep_gemm.hip): over all B*N token rows for
// the projections and weight gradients, batched per image for the N x N attention.  The element-wise and
// row-wise pieces between them are the small kernels below.  Gradients follow the chain rule of exactly
// this graph; there is no gradient with respect to the (frozen) tokens.
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

constexpr float LOG2E_F = 1.4426950408889634f;

// in-place softmax of every row (one wave per row)
__global__ __launch_bounds__(256) void ep_rowsoftmax_kernel(float* __restrict__ S, int64_t rows, int n) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float* row = S + r * n;
  float m = -INFINITY;
  for (int j = lane; j < n; j += 64) m = fmaxf(m, row[j]);
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < n; j += 64) l += __builtin_amdgcn_exp2f((row[j] - m) * LOG2E_F);
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int j = lane; j < n; j += 64) row[j] = __builtin_amdgcn_exp2f((row[j] - m) * LOG2E_F) * inv;
}

// dS <- A * (dS - sum_j A dS) per row
__global__ __launch_bounds__(256) void ep_rowsoftmax_bwd_kernel(const float* __restrict__ A, float* __restrict__ dS,
                                                              int64_t rows, int n) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* a = A + r * n;
  float* d = dS + r * n;
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s = fmaf(a[j], d[j], s);
  s = wave_sum(s);
  for (int j = lane; j < n; j += 64) d[j] = a[j] * (d[j] - s);
}

__global__ __launch_bounds__(256) void ep_tanh_kernel(float* __restrict__ H, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f4 v = reinterpret_cast<f4*>(H)[i];
  v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
  reinterpret_cast<f4*>(H)[i] = v;
}

// H (holding tanh output) <- dG = ds[row] * w2[col] * (1 - H^2)
__global__ __launch_bounds__(256) void ep_tanh_bwd_kernel(float* __restrict__ H, const float* __restrict__ ds,
                                                        const float* __restrict__ w2, int64_t rows, int D4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D4) return;
  const int64_t r = i / D4;
  const int c = (int)(i % D4);
  const f4 h = reinterpret_cast<f4*>(H)[i];
  const f4 w = reinterpret_cast<const f4*>(w2)[c];
  const float g = ds[r];
  reinterpret_cast<f4*>(H)[i] = g * w * (1.0f - h * h);
}

// out[r] = X[r,:] . V[(r / div),:] + bias   (one wave per row; div = rows -> one shared vector)
__global__ __launch_bounds__(256) void ep_rowdot_kernel(const float* __restrict__ X, const float* __restrict__ V,
                                                      const float* __restrict__ bias, int64_t rows, int64_t div, int D,
                                                      float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const f4* x = reinterpret_cast<const f4*>(X + r * D);
  const f4* v = reinterpret_cast<const f4*>(V + (r / div) * D);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane; c < D / 4; c += 64) acc += x[c] * v[c];
  const float s = wave_sum((acc.x + acc.y) + (acc.z + acc.w));
  if (lane == 0) out[r] = s + (bias ? bias[0] : 0.f);
}

// per image: a = softmax_n(s); out[b, d] = sum_n a[n] Xa[b,n,d].  grid (B, ceil(D/256))
__global__ __launch_bounds__(256) void ep_abmilp_pool_kernel(const float* __restrict__ s, const float* __restrict__ Xa,
                                                           int N, int D, float* __restrict__ a_out,
                                                           float* __restrict__ attn_map, float* __restrict__ out) {
  extern __shared__ float sa[];           // N floats
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sb = s + (int64_t)b * N;
  float m = -INFINITY;
  for (int n = tid; n < N; n += 256) m = fmaxf(m, sb[n]);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float l = 0.f;
  for (int n = tid; n < N; n += 256) { const float e = __builtin_amdgcn_exp2f((sb[n] - m) * LOG2E_F); sa[n] = e; l += e; }
  l = wave_sum(l);
  if ((tid & 63) == 0) red[tid >> 6] = l;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int n = tid; n < N; n += 256) {
    const float w = sa[n] * inv;
    sa[n] = w;
    if (blockIdx.y == 0) { a_out[(int64_t)b * N + n] = w; if (attn_map) attn_map[(int64_t)b * N + n] = w; }
  }
  __syncthreads();
  const int d = blockIdx.y * 256 + tid;
  if (d >= D) return;
  const float* xb = Xa + (int64_t)b * N * D + d;
  float acc0 = 0.f, acc1 = 0.f;
  int n = 0;
  for (; n + 1 < N; n += 2) { acc0 = fmaf(sa[n], xb[(int64_t)n * D], acc0); acc1 = fmaf(sa[n + 1], xb[(int64_t)(n + 1) * D], acc1); }
  if (n < N) acc0 = fmaf(sa[n], xb[(int64_t)n * D], acc0);
  out[(int64_t)b * D + d] = acc0 + acc1;
}

// per image: ds = a * (da - sum_n a da)
__global__ __launch_bounds__(256) void ep_abmilp_ds_kernel(const float* __restrict__ a, const float* __restrict__ da,
                                                         int N, float* __restrict__ ds) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int n = tid; n < N; n += 256) s = fmaf(a[(int64_t)b * N + n], da[(int64_t)b * N + n], s);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int n = tid; n < N; n += 256) ds[(int64_t)b * N + n] = a[(int64_t)b * N + n] * (da[(int64_t)b * N + n] - s);
}

// dXa[r, :] = a[r] * dout[r / N, :]
__global__ __launch_bounds__(256) void ep_outer_rows_kernel(const float* __restrict__ a, const float* __restrict__ dout,
                                                          int64_t rows, int N, int D4, float* __restrict__ dXa) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D4) return;
  const int64_t r = i / D4;
  const int c = (int)(i % D4);
  reinterpret_cast<f4*>(dXa)[i] = a[r] * reinterpret_cast<const f4*>(dout)[(r / N) * D4 + c];
}

// partial[rs][col] = sum over the rows of chunk rs of w[r] * src[r, col]   (w == nullptr: plain sum)
// grid (ceil(ncol/64), RS), 256 threads = 64 columns x 4 row lanes; fixed order -> reproducible
constexpr int WCS_RS = 64;
__global__ __launch_bounds__(256) void ep_wcolsum_kernel(const float* __restrict__ src, const float* __restrict__ w,
                                                       int64_t rows, int ncol, float* __restrict__ partial) {
  __shared__ float sm[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float s0 = 0.f, s1 = 0.f;
  if (col < ncol) {
    int64_t r = r0 + ty;
    for (; r + 4 < r1; r += 8) {
      s0 = fmaf(w ? w[r] : 1.f, src[r * ncol + col], s0);
      s1 = fmaf(w ? w[r + 4] : 1.f, src[(r + 4) * ncol + col], s1);
    }
    for (; r < r1; r += 4) s0 = fmaf(w ? w[r] : 1.f, src[r * ncol + col], s0);
  }
  sm[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && col < ncol)
    partial[(int64_t)blockIdx.y * ncol + col] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}

// out[0] (+)= sum_i v[i]   (one workgroup, fixed order; eight independent 16-byte loads per thread in flight: as a chain of
// dependent scalar loads the 65536 row gradients of a 256-image step took 60 us)
__global__ __launch_bounds__(256) void ep_sum_kernel(const float* __restrict__ v, int64_t n, int accumulate,
                                                   float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(v) % 16 == 0) ? n / 4 : 0;
  const f4* v4 = reinterpret_cast<const f4*>(v);
  int64_t i = threadIdx.x;
  for (; i + 7 * 256 < n4; i += 8 * 256) {
    f4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = v4[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (t[u].x + t[u].y) + (t[u].z + t[u].w);
  }
  for (; i < n4; i += 256) { const f4 t = v4[i]; s += (t.x + t.y) + (t.z + t.w); }
  for (int64_t j = 4 * n4 + threadIdx.x; j < n; j += 256) s += v[j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float t = (red[0] + red[1]) + (red[2] + red[3]); out[0] = accumulate ? out[0] + t : t; }
}

// The predictor's tanh layer backward in ONE read of H (round 6; before: a weighted column sum, the element-wise kernel and a
// plain column sum -- 1.2 GB of traffic per 256-image step at D = 1152 instead of 0.6):
//   pw2[rs][col] = sum_{r in chunk rs} ds[r] H[r, col]            (dw2 partials; H = tanh output)
//   H[r, col]   <- dG = ds[r] w2[col] (1 - H[r, col]^2)
//   pb1[rs][col] = sum_{r in chunk rs} dG[r, col]                 (db1 partials)
// grid (ceil(ncol/64), RS), 256 threads = 64 columns x 4 row lanes, the chunking and the summation order of ep_wcolsum_kernel.
__global__ __launch_bounds__(256) void ep_tanh_bwd_sums_kernel(float* __restrict__ H, const float* __restrict__ ds,
                                                             const float* __restrict__ w2, int64_t rows, int ncol,
                                                             float* __restrict__ pw2, float* __restrict__ pb1) {
  __shared__ float sm[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  if (col < ncol) {
    const float wc = w2[col];
    int64_t r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
      float h[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { h[u] = H[(r + 4 * u) * ncol + col]; g[u] = ds[r + 4 * u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dg = g[u] * wc * (1.0f - h[u] * h[u]);
        H[(r + 4 * u) * ncol + col] = dg;
        if (u & 1) { a1 = fmaf(g[u], h[u], a1); b1 += dg; } else { a0 = fmaf(g[u], h[u], a0); b0 += dg; }
      }
    }
    for (; r < r1; r += 4) {
      const float h = H[r * ncol + col], g = ds[r];
      const float dg = g * wc * (1.0f - h * h);
      H[r * ncol + col] = dg;
      a0 = fmaf(g, h, a0); b0 += dg;
    }
  }
  sm[0][ty][tx] = a0 + a1; sm[1][ty][tx] = b0 + b1;
  __syncthreads();
  if (ty == 0 && col < ncol) {
    pw2[(int64_t)blockIdx.y * ncol + col] = (sm[0][0][tx] + sm[0][1][tx]) + (sm[0][2][tx] + sm[0][3][tx]);
    pb1[(int64_t)blockIdx.y * ncol + col] = (sm[1][0][tx] + sm[1][1][tx]) + (sm[1][2][tx] + sm[1][3][tx]);
  }
}

// ---------------------------------------------------------------------------------------------
struct AbWs {
  float *QKV, *SA, *O, *Xa, *H, *s, *a, *da, *ds, *dXa, *dS, *dQKV, *part, *part2, *stage, *skws;
  size_t skws_floats;
  // contractions on the bf16-plane kernel (ep_planes.hip): planes of the three weight matrices (qkv natural; proj_w, w1 both
  // orientations)
  uint16_t *plQ, *plP, *plPT, *plW1, *plW1T;
  size_t pool_total;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

// EP_ABMILP_PLANES=0: every contraction on the f32 matrix instruction (rounds 1 - 2).  Default: the weight contractions of
// the forward pass (qkv, proj, the predictor's first layer) and the two activation gradients that go through a weight (dXa, dO)
// on the bf16 pipe at fp32 accuracy -- 1.22 of the step's 2.32 TFLOP at 256 x 1152, B = 256: 22.3 -> 20.8 ms per step (the
// kernel runs them at ~140 TFLOP/s against ~100 on the f32 instruction: on operands that stream from HBM its 64 x 128 tile is
// bound by the CU's fill path).
static bool ab_planes() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_ABMILP_PLANES"); on = e ? atoi(e) : 1; }
  return on != 0;
}

static int64_t ab_offsets(const ep_abmilp_dims& d, int64_t offs[9]) {
  const int64_t D = d.D;
  const int64_t sizes[9] = {3 * D * D, D * D, D, D * D, D, D, 1, (int64_t)d.C * D, d.C};
  int64_t off = 0;
  for (int i = 0; i < 9; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static AbWs ab_carve(const ep_abmilp_dims& d, void* base, bool head) {
  AbWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t BN = (size_t)d.B * d.N, D = d.D;
  w.QKV = take(BN * 3 * D); w.SA = take(BN * d.N); w.O = take(BN * D); w.Xa = take(BN * D); w.H = take(BN * D);
  w.s = take(BN); w.a = take(BN); w.da = take(BN); w.ds = take(BN);
  w.dXa = take(BN * D); w.dS = take(BN * d.N); w.dQKV = take(BN * 3 * D);
  w.part = take((size_t)WCS_RS * D); w.part2 = take((size_t)WCS_RS * D); w.stage = take(16 * D);
  w.skws_floats = (size_t)16 * D * D; w.skws = take(w.skws_floats);     // split-K slices of the D x D weight gradients
  if (ab_planes()) {
    auto take16 = [&](size_t n) { return reinterpret_cast<uint16_t*>(take((n + 1) / 2)); };
    w.plQ = take16(planes_elems(3 * d.D, d.D));
    w.plP = take16(planes_elems(d.D, d.D)); w.plPT = take16(planes_elems(d.D, d.D));
    w.plW1 = take16(planes_elems(d.D, d.D)); w.plW1T = take16(planes_elems(d.D, d.D));
  }
  w.pool_total = off;
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    const size_t B = d.B;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[9];
    w.opt_ws_bytes = optim_workspace_bytes(ab_offsets(d, offs), 9);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int ab_check(const ep_abmilp_dims& d, const void* x, int x_dtype, int64_t bstride, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0, EP_E_ARG, "abmilp dims must be positive");
  EP_REQUIRE(d.D % 4 == 0, EP_E_SHAPE, "abmilp: D = %d must be a multiple of 4", d.D);
  EP_REQUIRE((int64_t)d.B * d.N <= 2000000 && d.B <= 65535, EP_E_UNSUPPORTED, "abmilp: B*N = %lld exceeds one launch (split the batch)", (long long)d.B * d.N);
  EP_REQUIRE((size_t)d.N * 4 <= 60000, EP_E_UNSUPPORTED, "abmilp: N too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "abmilp head: C must be positive");
  if (x) {
    EP_REQUIRE(x_dtype == EP_DTYPE_F32, EP_E_UNSUPPORTED, "abmilp: fp32 tokens only");
    EP_REQUIRE(bstride == (int64_t)d.N * d.D, EP_E_SHAPE, "abmilp: tokens must be contiguous (batch stride %lld != N*D)", (long long)bstride);
    EP_REQUIRE(aligned16(x), EP_E_ALIGN, "abmilp: tokens must be 16-byte aligned");
  }
  return 0;
}

static int ab_params_ok(const ep_abmilp_params* p, const char* what) {
  EP_REQUIRE(p && p->qkv && p->proj_w && p->proj_b && p->w1 && p->b1 && p->w2 && p->b2, EP_E_ARG, "%s: null tensor", what);
  EP_REQUIRE(aligned16(p->qkv) && aligned16(p->proj_w) && aligned16(p->proj_b) && aligned16(p->w1) && aligned16(p->b1) &&
             aligned16(p->w2), EP_E_ALIGN, "%s: tensors must be 16-byte aligned", what);
  return 0;
}

static GemmParams mk(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

static int wcolsum(const float* src, const float* wgt, int64_t rows, int ncol, int accumulate, float* out, const AbWs& w,
                   hipStream_t st) {
  hipLaunchKernelGGL(ep_wcolsum_kernel, dim3((ncol + 63) / 64, WCS_RS), dim3(256), 0, st, src, wgt, rows, ncol, w.part);
  EP_LAUNCH_CHECK("ep_wcolsum_kernel");
  return reduce_partials(w.part, WCS_RS, ncol, 1.0f, accumulate, out, w.stage, st);
}

// C (M x N, ldc) (+)= A (M x K, lda) . (the rowsW x Kw matrix whose planes are given)^T + bias
static int ab_pl(const float* A, int64_t lda, const uint16_t* pl, int rowsW, int Kw, float* C, int64_t ldc, int M, int N, int K,
                 const float* bias, int accumulate, hipStream_t st) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.bias = bias; g.accumulate = accumulate;
  g.Bpl = pl; g.ldbp = (int64_t)round_up((size_t)Kw, 32); g.pl_term = (int64_t)rowsW * g.ldbp;
  return gemm_planes(g, 1, st);
}

static int ab_forward_core(const ep_abmilp_dims& d, const float* x, const ep_abmilp_params& pr, const AbWs& w,
                           float* out, float* attn_map, hipStream_t st) {
  const int D = d.D, N = d.N, BN = d.B * d.N;
  const float scale = (float)pow((double)D, -0.5);                       // head_dim ** -0.5, one head (models_vit.py:59-60)
  const bool pl = w.plQ != nullptr;
  if (pl) {
    PlaneSpec sp[3] = {{pr.qkv, 3 * D, D, D, w.plQ, nullptr}, {pr.proj_w, D, D, D, w.plP, w.plPT}, {pr.w1, D, D, D, w.plW1, w.plW1T}};
    EP_TRY(planes_split(sp, 3, st));
    EP_TRY(ab_pl(x, D, w.plQ, 3 * D, D, w.QKV, 3 * D, BN, 3 * D, D, nullptr, 0, st));
  } else {
    EP_TRY(gemm(true, true, mk(x, D, pr.qkv, D, w.QKV, 3 * D, BN, 3 * D, D), 1, st));
  }
  {
    GemmParams g = mk(w.QKV, 3 * D, w.QKV + D, 3 * D, w.SA, N, N, N, D);   // S = (q scale) k^T per image
    g.sAz = (int64_t)N * 3 * D; g.sBz = g.sAz; g.sCz = (int64_t)N * N; g.alpha = scale;
    EP_TRY(gemm(true, true, g, d.B, st));
  }
  hipLaunchKernelGGL(ep_rowsoftmax_kernel, dim3((BN + 3) / 4), dim3(256), 0, st, w.SA, (int64_t)BN, N);
  {
    GemmParams g = mk(w.SA, N, w.QKV + 2 * D, 3 * D, w.O, D, N, D, N);     // O = A v per image
    g.sAz = (int64_t)N * N; g.sBz = (int64_t)N * 3 * D; g.sCz = (int64_t)N * D; g.extB = D;
    EP_TRY(gemm(true, false, g, d.B, st));
  }
  if (pl) {
    EP_TRY(ab_pl(w.O, D, w.plP, D, D, w.Xa, D, BN, D, D, pr.proj_b, 0, st));
    EP_TRY(ab_pl(w.Xa, D, w.plW1, D, D, w.H, D, BN, D, D, pr.b1, 0, st));
  } else {
    {
      GemmParams g = mk(w.O, D, pr.proj_w, D, w.Xa, D, BN, D, D); g.bias = pr.proj_b;
      EP_TRY(gemm(true, true, g, 1, st));
    }
    {
      GemmParams g = mk(w.Xa, D, pr.w1, D, w.H, D, BN, D, D); g.bias = pr.b1;
      EP_TRY(gemm(true, true, g, 1, st));
    }
  }
  const int64_t n4 = (int64_t)BN * D / 4;
  hipLaunchKernelGGL(ep_tanh_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.H, n4);
  hipLaunchKernelGGL(ep_rowdot_kernel, dim3((BN + 3) / 4), dim3(256), 0, st, w.H, pr.w2, pr.b2, (int64_t)BN, (int64_t)BN, D, w.s);
  hipLaunchKernelGGL(ep_abmilp_pool_kernel, dim3(d.B, (D + 255) / 256), dim3(256), (size_t)N * 4, st, w.s, w.Xa, N, D, w.a,
                     attn_map, out);
  EP_LAUNCH_CHECK("abmilp forward kernels");
  return 0;
}

static int ab_backward_core(const ep_abmilp_dims& d, const float* x, const ep_abmilp_params& pr, const float* dout,
                            const ep_abmilp_params& gr, int acc, const AbWs& w, hipStream_t st) {
  const int D = d.D, N = d.N, BN = d.B * d.N;
  const float scale = (float)pow((double)D, -0.5);
  const int64_t n4 = (int64_t)BN * D / 4;
  const unsigned eg = (unsigned)((n4 + 255) / 256);
  // pooling + predictor
  hipLaunchKernelGGL(ep_rowdot_kernel, dim3((BN + 3) / 4), dim3(256), 0, st, w.Xa, dout, (const float*)nullptr, (int64_t)BN,
                     (int64_t)N, D, w.da);
  hipLaunchKernelGGL(ep_abmilp_ds_kernel, dim3(d.B), dim3(256), 0, st, w.a, w.da, N, w.ds);
  EP_LAUNCH_CHECK("abmilp backward kernels (1)");
  hipLaunchKernelGGL(ep_sum_kernel, dim3(1), dim3(256), 0, st, w.ds, (int64_t)BN, acc, gr.b2);
  static int fused_tanh = -1;                // EP_ABMILP_TANH_FUSED=0: the three kernels of rounds 1 - 5
  if (fused_tanh < 0) { const char* e = getenv("EP_ABMILP_TANH_FUSED"); fused_tanh = e ? atoi(e) : 1; }
  if (fused_tanh) {
    // dw2 = sum_r ds[r] H[r,:], H <- dG, db1 = sum_r dG[r,:] in one read of H
    hipLaunchKernelGGL(ep_tanh_bwd_sums_kernel, dim3((D + 63) / 64, WCS_RS), dim3(256), 0, st, w.H, w.ds, pr.w2, (int64_t)BN, D, w.part, w.part2);
    EP_LAUNCH_CHECK("ep_tanh_bwd_sums_kernel");
    EP_TRY(reduce_partials(w.part, WCS_RS, D, 1.0f, acc, gr.w2, w.stage, st));
    EP_TRY(reduce_partials(w.part2, WCS_RS, D, 1.0f, acc, gr.b1, w.stage, st));
  } else {
    EP_TRY(wcolsum(w.H, w.ds, BN, D, acc, gr.w2, w, st));                            // dw2 = sum_r ds[r] H[r,:]
    hipLaunchKernelGGL(ep_tanh_bwd_kernel, dim3(eg), dim3(256), 0, st, w.H, w.ds, pr.w2, (int64_t)BN, D / 4);   // H <- dG
    EP_LAUNCH_CHECK("abmilp backward kernels (2)");
    EP_TRY(wcolsum(w.H, nullptr, BN, D, acc, gr.b1, w, st));
  }
  {
    GemmParams g = mk(w.H, D, w.Xa, D, gr.w1, D, D, D, BN); g.accumulate = acc;       // dW1 = dG^T Xa
    g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(false, false, g, 1, st));
  }
  hipLaunchKernelGGL(ep_outer_rows_kernel, dim3(eg), dim3(256), 0, st, w.a, dout, (int64_t)BN, N, D / 4, w.dXa);
  EP_LAUNCH_CHECK("ep_outer_rows_kernel");
  const bool pl = w.plQ != nullptr;          // (the planes are those the forward pass of this step split: same weights)
  if (pl) {
    EP_TRY(ab_pl(w.H, D, w.plW1T, D, D, w.dXa, D, BN, D, D, nullptr, 1, st));         // dXa += dG W1
  } else {
    GemmParams g = mk(w.H, D, pr.w1, D, w.dXa, D, BN, D, D); g.accumulate = 1;        // dXa += dG W1
    EP_TRY(gemm(true, false, g, 1, st));
  }
  // output projection of the self-attention
  EP_TRY(wcolsum(w.dXa, nullptr, BN, D, acc, gr.proj_b, w, st));
  {
    GemmParams g = mk(w.dXa, D, w.O, D, gr.proj_w, D, D, D, BN); g.accumulate = acc;  // dWp = dXa^T O
    g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(false, false, g, 1, st));
  }
  float* dO = w.H;                                                                   // dG is dead from here on
  if (pl) EP_TRY(ab_pl(w.dXa, D, w.plPT, D, D, dO, D, BN, D, D, nullptr, 0, st));    // dO = dXa Wp
  else EP_TRY(gemm(true, false, mk(w.dXa, D, pr.proj_w, D, dO, D, BN, D, D), 1, st));
  // attention
  {
    GemmParams g = mk(dO, D, w.QKV + 2 * D, 3 * D, w.dS, N, N, N, D);                 // dA = dO v^T
    g.sAz = (int64_t)N * D; g.sBz = (int64_t)N * 3 * D; g.sCz = (int64_t)N * N;
    EP_TRY(gemm(true, true, g, d.B, st));
  }
  {
    GemmParams g = mk(w.SA, N, dO, D, w.dQKV + 2 * D, 3 * D, N, D, N);               // dv = A^T dO
    g.sAz = (int64_t)N * N; g.sBz = (int64_t)N * D; g.sCz = (int64_t)N * 3 * D;
    EP_TRY(gemm(false, false, g, d.B, st));
  }
  hipLaunchKernelGGL(ep_rowsoftmax_bwd_kernel, dim3((BN + 3) / 4), dim3(256), 0, st, w.SA, w.dS, (int64_t)BN, N);
  EP_LAUNCH_CHECK("ep_rowsoftmax_bwd_kernel");
  {
    GemmParams g = mk(w.dS, N, w.QKV + D, 3 * D, w.dQKV, 3 * D, N, D, N);             // dq = scale dS k
    g.sAz = (int64_t)N * N; g.sBz = (int64_t)N * 3 * D; g.sCz = g.sBz; g.extB = D; g.alpha = scale;
    EP_TRY(gemm(true, false, g, d.B, st));
  }
  {
    GemmParams g = mk(w.dS, N, w.QKV, 3 * D, w.dQKV + D, 3 * D, N, D, N);             // dk = scale dS^T q
    g.sAz = (int64_t)N * N; g.sBz = (int64_t)N * 3 * D; g.sCz = g.sBz; g.extB = D; g.alpha = scale;
    EP_TRY(gemm(false, false, g, d.B, st));
  }
  {
    // (On the planes kernel -- dQKV transposed in fp32, the tokens split into planes of their transpose -- this contraction
    // over the B N = 65536 token rows takes 4.75 ms + 0.48 ms of transpose and split against 5.1 ms here: both stream 40 GB
    // of operand panels; not kept.)
    GemmParams g = mk(w.dQKV, 3 * D, x, D, gr.qkv, D, 3 * D, D, BN); g.accumulate = acc;   // dWqkv = dQKV^T x
    g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(false, false, g, 1, st));
  }
  return 0;
}

static ep_abmilp_params ab_views(float* base, const int64_t o[9]) {
  ep_abmilp_params p;
  p.qkv = base + o[0]; p.proj_w = base + o[1]; p.proj_b = base + o[2]; p.w1 = base + o[3]; p.b1 = base + o[4];
  p.w2 = base + o[5]; p.b2 = base + o[6];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

size_t ep_abmilp_pool_workspace_bytes(const ep_abmilp_dims* dims) {
  if (!dims || ab_check(*dims, nullptr, 0, 0, false) != 0) return 0;
  return ab_carve(*dims, nullptr, false).total;
}

int ep_abmilp_pool_forward(const ep_abmilp_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                           const ep_abmilp_params* params, float* out, float* attn_map, void* ws, size_t ws_bytes,
                           ep_stream_t stream) {
  EP_REQUIRE(dims && x && out && ws, EP_E_ARG, "ep_abmilp_pool_forward: null pointer");
  EP_TRY(ab_check(*dims, x, x_dtype, x_bstride, false));
  EP_TRY(ab_params_ok(params, "ep_abmilp_pool_forward"));
  EP_REQUIRE(aligned16(ws) && aligned16(out), EP_E_ALIGN, "ep_abmilp_pool_forward: out / ws must be 16-byte aligned");
  const AbWs w = ab_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_abmilp_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return ab_forward_core(*dims, static_cast<const float*>(x), *params, w, out, attn_map, (hipStream_t)stream);
}

int ep_abmilp_pool_backward(const ep_abmilp_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const ep_abmilp_params* params, const float* dout, const ep_abmilp_params* grads,
                            int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && dout && ws, EP_E_ARG, "ep_abmilp_pool_backward: null pointer");
  EP_TRY(ab_check(*dims, x, x_dtype, x_bstride, false));
  EP_TRY(ab_params_ok(params, "ep_abmilp_pool_backward(params)"));
  EP_TRY(ab_params_ok(grads, "ep_abmilp_pool_backward(grads)"));
  EP_REQUIRE(aligned16(ws) && aligned16(dout), EP_E_ALIGN, "ep_abmilp_pool_backward: dout / ws must be 16-byte aligned");
  const AbWs w = ab_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_abmilp_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return ab_backward_core(*dims, static_cast<const float*>(x), *params, dout, *grads, accumulate, w, (hipStream_t)stream);
}

int64_t ep_abmilp_head_param_offsets(const ep_abmilp_dims* dims, int64_t offsets[9]) { return ab_offsets(*dims, offsets); }

size_t ep_abmilp_head_workspace_bytes(const ep_abmilp_dims* dims) {
  if (!dims || ab_check(*dims, nullptr, 0, 0, true) != 0) return 0;
  return ab_carve(*dims, nullptr, true).total;
}

int64_t ep_abmilp_head_workspace_logits_offset(const ep_abmilp_dims* dims, int32_t* ldl) {
  if (!dims || ab_check(*dims, nullptr, 0, 0, true) != 0) return -1;
  char* base = reinterpret_cast<char*>(uintptr_t(1) << 20);   // ab_carve() only does address arithmetic on a non-null base
  const AbWs w = ab_carve(*dims, base, true);
  if (ldl) *ldl = w.ldl;
  return reinterpret_cast<char*>(w.logits) - base;
}

int ep_abmilp_head_train_step(const ep_abmilp_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_abmilp_head_train_step: null pointer");
  const ep_abmilp_dims& d = s->dims;
  EP_TRY(ab_check(d, (s->phases & 1) ? s->x : nullptr, s->x_dtype, s->x_bstride, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const AbWs w = ab_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_abmilp_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  EP_REQUIRE(s->arith == EP_ARITH_F32 || s->arith == EP_ARITH_BF16_AUTOCAST, EP_E_ARG, "ep_abmilp_head_train_step: arith %d", s->arith);
  const ArithScope arith_scope(s->arith);            // AMP-bf16: every contraction below as one bf16 product (ep_gemm.hip: gemm_b3_ok, ep_planes.hip)
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[9];
  const int64_t total = ab_offsets(d, offs);
  const ep_abmilp_params pr = ab_views(s->params, offs), gr = ab_views(s->grads, offs);
  float* Wc = s->params + offs[7]; float* bc = s->params + offs[8];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    const float* x = static_cast<const float*>(s->x);
    EP_TRY(ab_forward_core(d, x, pr, w, w.y, nullptr, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, s->grads + offs[7], s->grads + offs[8],
                           s->accumulate, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    EP_TRY(ab_backward_core(d, x, pr, w.dy, gr, s->accumulate, w, st));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t D = d.D;
    const int64_t sizes[9] = {3 * D * D, D * D, D, D * D, D, D, 1, (int64_t)d.C * D, d.C};
    const int trust[9] = {1, 1, 0, 1, 0, 1, 0, 1, 0};                 // ndim > 1 (util/lars.py:22); w2 is (1, D)
    ep_segment segs[9];
    for (int i = 0; i < 9; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? 9 : 0, s->lr, s->weight_decay,
                      s->momentum, s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step,
                      s->found_inf, s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_abmilp_head_eval_forward(const ep_abmilp_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                                const float* params, const float* running_mean, const float* running_var, float bn_eps,
                                float* logits, int ldl, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_abmilp_head_eval_forward: null pointer");
  const ep_abmilp_dims& d = *dims;
  EP_TRY(ab_check(d, x, x_dtype, x_bstride, true));
  const AbWs w = ab_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_abmilp_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[9];
  ab_offsets(d, offs);
  const ep_abmilp_params pr = ab_views(const_cast<float*>(params), offs);
  EP_TRY(ab_forward_core(d, static_cast<const float*>(x), pr, w, w.y, nullptr, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[7], params + offs[8], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
