// DINOv2-block pooling (reference poolings/other_pool.py:299-318 DinoViTBlockPooling with poolings/dinov2_layers/block.py:43-113
// Block, attention.py:37-69 Attention, mlp.py Mlp; registry entry probe_heads.py:80: DinoViTBlockPooling(d_model=dim) -> 8
// heads, no qkv bias, LayerNorm eps 1e-5, GELU MLP x4, no LayerScale, no drop path):
//     x1 = x + proj(MHSA(norm1(x))) ;  x2 = x1 + fc2(gelu(fc1(norm2(x1)))) ;  out[b] = mean_n x2[b,n]
// A whole transformer block over every token: 24 N D^2 + 4 N^2 D FLOP per image forward (3.8 GFLOP at 256 x 768) and twice
// that backward -- the matrix-core-bound end of the head family, like AbMILP.  Every contraction is one call of the exact-fp32
// MFMA kernel (ep_gemm.hip): over all B N token rows for the projections / MLP and their weight gradients, batched per image
// and per head for the N x N attention.  The mean over the tokens is used where it commutes: the gradient of x2 is the same
// row dout[b] / N for every token, so d fc2.weight = (dout / N)^T (sum_n h1) is a (D x 4D x B) contraction and d h1 needs
// one (B x 4D x D) contraction instead of a (B N)-row one.  No gradient with respect to the (frozen) tokens: the first
// LayerNorm's backward stops at its affine parameters.
#include "ep_side.h"
#include "ep_headkernels.h"

namespace ep {

constexpr float DV_LOG2E = 1.4426950408889634f;

// in-place softmax of every row (one wave per row)
__global__ __launch_bounds__(256) void ep_dv_softmax_kernel(float* __restrict__ S, int64_t rows, int n) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float* row = S + r * n;
  float m = -INFINITY;
  for (int j = lane; j < n; j += 64) m = fmaxf(m, row[j]);
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < n; j += 64) l += __builtin_amdgcn_exp2f((row[j] - m) * DV_LOG2E);
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int j = lane; j < n; j += 64) row[j] = __builtin_amdgcn_exp2f((row[j] - m) * DV_LOG2E) * inv;
}
// dS <- A * (dS - sum_j A dS) per row
__global__ __launch_bounds__(256) void ep_dv_softmax_bwd_kernel(const float* __restrict__ A, float* __restrict__ dS, int64_t rows,
                                                              int n) {
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

// per image column sums over the N token rows: out[b,c] = alpha * sum_n src[b,n,c]     (a thread owns 4 columns)
__global__ __launch_bounds__(256) void ep_dv_imgsum_kernel(const float* __restrict__ src, int N, int W, float alpha,
                                                         float* __restrict__ out) {
  const int b = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= W) return;
  const float* p = src + (int64_t)b * N * W + c;
  f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int n = 0; n < N; ++n) s += *reinterpret_cast<const f4*>(p + (int64_t)n * W);
  *reinterpret_cast<f4*>(out + (int64_t)b * W + c) = s * alpha;
}

// dst[b,n,:] = alpha * src[b,:]      (broadcast of a per-image row over its N tokens; total = B N W / 4)
__global__ void ep_dv_bcast_kernel(const float* __restrict__ src, int64_t total4, int N, int W4, float alpha, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int64_t b = i / ((int64_t)N * W4); const int c = (int)(i % W4);
  reinterpret_cast<f4*>(dst)[i] = reinterpret_cast<const f4*>(src)[b * W4 + c] * alpha;
}
// dpre[b,n,:] = g[b,:] * gelu'(pre[b,n,:])     (the upstream gradient of h1 is the same row for every token of an image)
__global__ void ep_dv_gelu_bwd_bcast_kernel(const float* __restrict__ pre, const float* __restrict__ g, int64_t total4, int N, int W4,
                                            float* __restrict__ dpre) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int64_t b = i / ((int64_t)N * W4); const int c = (int)(i % W4);
  const f4 v = reinterpret_cast<const f4*>(pre)[i];
  const f4 u = reinterpret_cast<const f4*>(g)[b * W4 + c];
  auto dg = [](float x) { return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x); };
  reinterpret_cast<f4*>(dpre)[i] = f4{u.x * dg(v.x), u.y * dg(v.y), u.z * dg(v.z), u.w * dg(v.w)};
}

// Column reductions over MANY rows in two deterministic stages: grid (ceil(W / 64), RS); a workgroup = 64 columns x 4 row lanes
// over its row chunk -> part[which][rs][col]; the partials are summed by ep_reduce_partials_kernel.
//   XHAT = false: part[0][rs][c] = sum_r a[r,c]
//   XHAT = true : part[0][rs][c] = sum_r a[r,c] xhat[r,c] ; part[1][rs][c] = sum_r a[r,c]     (xhat from x and {mean, rstd})
template <bool XHAT>
__global__ __launch_bounds__(256) void ep_dv_colpart_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                          const float* __restrict__ stats, int64_t rows, int W,
                                                          float* __restrict__ part) {
  __shared__ float p0[4][64], p1[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float s0 = 0.f, s1 = 0.f;
  if (c < W)
    for (int64_t r = r0 + ty; r < r1; r += 4) {
      const float v = a[r * W + c];
      if (XHAT) { s0 = fmaf(v, (x[r * W + c] - stats[2 * r]) * stats[2 * r + 1], s0); s1 += v; }
      else s0 += v;
    }
  p0[ty][tx] = s0; p1[ty][tx] = s1;
  __syncthreads();
  if (ty == 0 && c < W) {
    part[(int64_t)blockIdx.y * W + c] = (p0[0][tx] + p0[1][tx]) + (p0[2][tx] + p0[3][tx]);
    if (XHAT) part[((int64_t)gridDim.y + blockIdx.y) * W + c] = (p1[0][tx] + p1[1][tx]) + (p1[2][tx] + p1[3][tx]);
  }
}

constexpr int DV_RS = 128;
// out0[c] (+)= sum_r a[r,c] (xhat[r,c] if x) ; out1[c] (+)= sum_r a[r,c] (only with x)
static int dv_colsum(const float* a, const float* x, const float* stats, int64_t rows, int W, int acc, float* out0, float* out1,
                     float* scratch, hipStream_t st) {
  const int nw = x ? 2 : 1;
  int rs = (int)((rows + 255) / 256);
  rs = rs < 1 ? 1 : (rs > DV_RS ? DV_RS : rs);
  const dim3 grid((W + 63) / 64, rs);
  if (x) hipLaunchKernelGGL(ep_dv_colpart_kernel<true>, grid, dim3(256), 0, st, a, x, stats, rows, W, scratch);
  else hipLaunchKernelGGL(ep_dv_colpart_kernel<false>, grid, dim3(256), 0, st, a, x, stats, rows, W, scratch);
  EP_LAUNCH_CHECK("ep_dv_colpart_kernel");
  float* stage = scratch + (size_t)rs * nw * W;
  EP_TRY(reduce_partials(scratch, rs, W, 1.0f, acc, out0, stage, st));
  if (x) EP_TRY(reduce_partials(scratch + (size_t)rs * W, rs, W, 1.0f, acc, out1, stage, st));
  return 0;
}

// ---------------------------------------------------------------------------------------------
constexpr int DV_NT = 13;     // n1.w n1.b | qkv.w | proj.w proj.b | n2.w n2.b | fc1.w fc1.b | fc2.w fc2.b | fc.weight fc.bias
struct DvWs {
  float *stat1, *h, *QKV, *S, *ctx, *x1, *stat2, *h2, *pre, *h1, *x2, *h1sum;
  float *g0, *g1, *dx1, *dh2, *dctx, *dS, *dQKV, *scr, *skws;
  size_t skws_floats;
  // contractions through a weight matrix on the bf16-plane kernel (ep_planes.hip): planes of qkv / proj / fc1 (both
  // orientations) and fc2
  uint16_t *plQ, *plQT, *plP, *plPT, *plF1, *plF1T, *plF2;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

// EP_DINOVIT_PLANES=0: every contraction on the f32 matrix instruction (rounds 1 - 2).  Default: the four forward projections
// and the three activation gradients that go through a weight on the bf16 pipe at fp32 accuracy (as in ep_abmilp.hip).
static bool dv_planes() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_DINOVIT_PLANES"); on = e ? atoi(e) : 1; }
  return on != 0;
}
// C (M x N, ldc) (+)= A (M x K, lda) . (the rowsW x Kw matrix whose planes are given)^T + bias
static int dv_pl(const float* A, int64_t lda, const uint16_t* pl, int rowsW, int Kw, float* C, int64_t ldc, int M, int N, int K,
                 const float* bias, int accumulate, hipStream_t st) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.bias = bias; g.accumulate = accumulate;
  g.Bpl = pl; g.ldbp = (int64_t)round_up((size_t)Kw, 32); g.pl_term = (int64_t)rowsW * g.ldbp;
  return gemm_planes(g, 1, st);
}

static void dv_sizes(const ep_dinovit_dims& d, int64_t sizes[DV_NT]) {
  const int64_t D = d.D, Hd = d.hidden;
  const int64_t s[DV_NT] = {D, D, 3 * D * D, D * D, D, D, D, Hd * D, Hd, D * Hd, D, (int64_t)d.C * D, d.C};
  for (int i = 0; i < DV_NT; ++i) sizes[i] = s[i];
}
static int64_t dv_offsets(const ep_dinovit_dims& d, int64_t offs[DV_NT]) {
  int64_t sizes[DV_NT];
  dv_sizes(d, sizes);
  int64_t off = 0;
  for (int i = 0; i < DV_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static DvWs dv_carve(const ep_dinovit_dims& d, void* base, bool head) {
  DvWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, N = d.N, Hd = d.hidden, R = B * N, H = d.H;
  w.stat1 = take(R * 2); w.h = take(R * D); w.QKV = take(R * 3 * D); w.S = take(B * H * N * N); w.ctx = take(R * D);
  w.x1 = take(R * D); w.stat2 = take(R * 2); w.h2 = take(R * D); w.pre = take(R * Hd); w.h1 = take(R * Hd); w.x2 = take(R * D);
  w.h1sum = take(B * Hd);
  w.g0 = take(B * D); w.g1 = take(B * Hd); w.dx1 = take(R * D); w.dh2 = take(R * D); w.dctx = take(R * D);
  w.dS = take(B * H * N * N); w.dQKV = take(R * 3 * D);
  const size_t wmax = 3 * D > Hd ? 3 * D : Hd;
  w.scr = take((size_t)(DV_RS + 16 + 2) * 2 * wmax);
  w.skws_floats = (size_t)16 * D * D; w.skws = take(w.skws_floats);      // split-K slices of the weight gradients
  if (dv_planes()) {
    auto take16 = [&](size_t n) { return reinterpret_cast<uint16_t*>(take((n + 1) / 2)); };
    w.plQ = take16(planes_elems(3 * d.D, d.D)); w.plQT = take16(planes_elems(d.D, 3 * d.D));
    w.plP = take16(planes_elems(d.D, d.D)); w.plPT = take16(planes_elems(d.D, d.D));
    w.plF1 = take16(planes_elems(d.hidden, d.D)); w.plF1T = take16(planes_elems(d.D, d.hidden));
    w.plF2 = take16(planes_elems(d.D, d.hidden));
  }
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[DV_NT];
    w.opt_ws_bytes = optim_workspace_bytes(dv_offsets(d, offs), DV_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int dv_check(const ep_dinovit_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0 && d.hidden > 0, EP_E_ARG, "dinovit dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0 && d.hidden % 4 == 0, EP_E_SHAPE,
             "dinovit: D %% H == 0 and D/H, D, hidden multiples of 4 (D=%d H=%d hidden=%d)", d.D, d.H, d.hidden);
  EP_REQUIRE((int64_t)d.B * d.N < (1ll << 31) / 4, EP_E_SHAPE, "dinovit: B * N too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "dinovit head: C must be positive");
  return 0;
}

static int dv_params_ok(const ep_dinovit_params* p, const char* what) {
  EP_REQUIRE(p, EP_E_ARG, "%s: null parameter struct", what);
  const float* ts[] = {p->n1_w, p->n1_b, p->qkv_w, p->proj_w, p->proj_b, p->n2_w, p->n2_b, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b};
  for (const float* t : ts) EP_REQUIRE(t && aligned16(t), EP_E_ALIGN, "%s: tensors must be non-null and 16-byte aligned", what);
  return 0;
}

static GemmParams vg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

// x: (B*N, D) fp32, contiguous
static int dv_forward_core(const ep_dinovit_dims& d, const float* x, const ep_dinovit_params& pr, const DvWs& w, float* out,
                           hipStream_t st) {
  const int D = d.D, N = d.N, B = d.B, H = d.H, dh = D / H, Hd = d.hidden, R = B * N;
  const float scale = (float)pow((double)dh, -0.5);                        // attention.py:49
  const int64_t nd = (int64_t)R * D;
  const unsigned eg = (unsigned)((nd + 255) / 256);
  EP_TRY(token_stats(x, 0, (int64_t)N * D, B, N, D, d.ln_eps, w.stat1, st));
  hipLaunchKernelGGL(ep_rowln_apply_kernel, dim3(eg), dim3(256), 0, st, x, w.stat1, pr.n1_w, pr.n1_b, nd, D, w.h);
  EP_LAUNCH_CHECK("ep_rowln_apply_kernel");
  const bool pl = w.plQ != nullptr;
  if (pl) {
    PlaneSpec sp[4] = {{pr.qkv_w, 3 * D, D, D, w.plQ, w.plQT}, {pr.proj_w, D, D, D, w.plP, w.plPT},
                       {pr.fc1_w, Hd, D, D, w.plF1, w.plF1T}, {pr.fc2_w, D, Hd, Hd, w.plF2, nullptr}};
    EP_TRY(planes_split(sp, 4, st));
    EP_TRY(dv_pl(w.h, D, w.plQ, 3 * D, D, w.QKV, 3 * D, R, 3 * D, D, nullptr, 0, st));                 // QKV = h Wqkv^T
  } else {
    EP_TRY(gemm(true, true, vg(w.h, D, pr.qkv_w, D, w.QKV, 3 * D, R, 3 * D, D), 1, st));
  }
  const int64_t s3 = (int64_t)N * 3 * D, sS = (int64_t)H * N * N;
  for (int hh = 0; hh < H; ++hh) {                                                                      // S_h = scale q_h k_h^T
    GemmParams g = vg(w.QKV + hh * dh, 3 * D, w.QKV + D + hh * dh, 3 * D, w.S + (int64_t)hh * N * N, N, N, N, dh);
    g.sAz = s3; g.sBz = s3; g.sCz = sS; g.alpha = scale;
    EP_TRY(gemm(true, true, g, B, st));
  }
  hipLaunchKernelGGL(ep_dv_softmax_kernel, dim3((unsigned)(((int64_t)R * H + 3) / 4)), dim3(256), 0, st, w.S, (int64_t)R * H, N);
  EP_LAUNCH_CHECK("ep_dv_softmax_kernel");
  for (int hh = 0; hh < H; ++hh) {                                                                      // ctx_h = A_h v_h
    GemmParams g = vg(w.S + (int64_t)hh * N * N, N, w.QKV + 2 * D + hh * dh, 3 * D, w.ctx + hh * dh, D, N, dh, N);
    g.sAz = sS; g.sBz = s3; g.sCz = (int64_t)N * D; g.extB = dh;
    EP_TRY(gemm(true, false, g, B, st));
  }
  EP_HIP(hipMemcpyAsync(w.x1, x, (size_t)nd * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (pl) EP_TRY(dv_pl(w.ctx, D, w.plP, D, D, w.x1, D, R, D, D, pr.proj_b, 1, st));
  else { GemmParams g = vg(w.ctx, D, pr.proj_w, D, w.x1, D, R, D, D); g.bias = pr.proj_b; g.accumulate = 1; EP_TRY(gemm(true, true, g, 1, st)); }
  EP_TRY(token_stats(w.x1, 0, (int64_t)N * D, B, N, D, d.ln_eps, w.stat2, st));
  hipLaunchKernelGGL(ep_rowln_apply_kernel, dim3(eg), dim3(256), 0, st, w.x1, w.stat2, pr.n2_w, pr.n2_b, nd, D, w.h2);
  if (pl) EP_TRY(dv_pl(w.h2, D, w.plF1, Hd, D, w.pre, Hd, R, Hd, D, pr.fc1_b, 0, st));
  else { GemmParams g = vg(w.h2, D, pr.fc1_w, D, w.pre, Hd, R, Hd, D); g.bias = pr.fc1_b; EP_TRY(gemm(true, true, g, 1, st)); }
  const int64_t n4 = (int64_t)R * Hd / 4;
  hipLaunchKernelGGL(ep_gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.h1);
  EP_LAUNCH_CHECK("ep_dinovit MLP kernels");
  EP_HIP(hipMemcpyAsync(w.x2, w.x1, (size_t)nd * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (pl) EP_TRY(dv_pl(w.h1, Hd, w.plF2, D, Hd, w.x2, D, R, D, Hd, pr.fc2_b, 1, st));
  else { GemmParams g = vg(w.h1, Hd, pr.fc2_w, Hd, w.x2, D, R, D, Hd); g.bias = pr.fc2_b; g.accumulate = 1; EP_TRY(gemm(true, true, g, 1, st)); }
  hipLaunchKernelGGL(ep_dv_imgsum_kernel, dim3(B, (D / 4 + 255) / 256), dim3(256), 0, st, w.x2, N, D, 1.0f / (float)N, out);
  hipLaunchKernelGGL(ep_dv_imgsum_kernel, dim3(B, (Hd / 4 + 255) / 256), dim3(256), 0, st, w.h1, N, Hd, 1.0f, w.h1sum);
  EP_LAUNCH_CHECK("ep_dv_imgsum_kernel");
  return 0;
}

static int dv_backward_core(const ep_dinovit_dims& d, const float* x, const ep_dinovit_params& pr, const float* dout,
                            const ep_dinovit_params& gr, int acc, const DvWs& w, hipStream_t st) {
  const int D = d.D, N = d.N, B = d.B, H = d.H, dh = D / H, Hd = d.hidden, R = B * N;
  const float scale = (float)pow((double)dh, -0.5);
  const int64_t nd = (int64_t)R * D, n4h = (int64_t)R * Hd / 4;
  const float invN = 1.0f / (float)N;
  // out[b] = mean_n x2[b,n]: d x2[b,n,:] = g0[b,:] = dout[b,:] / N for every token
  hipLaunchKernelGGL(ep_dv_bcast_kernel, dim3((unsigned)(((int64_t)B * D / 4 + 255) / 256)), dim3(256), 0, st, dout, (int64_t)B * D / 4,
                     1, D / 4, invN, w.g0);
  EP_LAUNCH_CHECK("ep_dv_bcast_kernel");
  // x2 = x1 + h1 W2^T + b2
  EP_TRY(colsum(dout, B, D, D, acc, gr.fc2_b, st));                                                      // sum_{b,n} g0 = sum_b dout
  { GemmParams g = vg(w.g0, D, w.h1sum, Hd, gr.fc2_w, Hd, D, Hd, B); g.accumulate = acc; EP_TRY(gemm(false, false, g, 1, st)); }   // d W2
  EP_TRY(gemm(true, false, vg(w.g0, D, pr.fc2_w, Hd, w.g1, Hd, B, Hd, D), 1, st));                        // g1 = g0 W2 (per image)
  hipLaunchKernelGGL(ep_dv_gelu_bwd_bcast_kernel, dim3((unsigned)((n4h + 255) / 256)), dim3(256), 0, st, w.pre, w.g1, n4h, N, Hd / 4,
                     w.h1);                                                                               // h1 <- dpre
  EP_LAUNCH_CHECK("ep_dv_gelu_bwd_bcast_kernel");
  float* dpre = w.h1;
  EP_TRY(dv_colsum(dpre, nullptr, nullptr, R, Hd, acc, gr.fc1_b, nullptr, w.scr, st));
  { GemmParams g = vg(dpre, Hd, w.h2, D, gr.fc1_w, D, Hd, D, R); g.accumulate = acc; EP_TRY(gemm(false, false, g, 1, st)); }       // d W1
  const bool pl = w.plQ != nullptr;            // (the planes are those the forward pass of this step split: same weights)
  if (pl) EP_TRY(dv_pl(dpre, Hd, w.plF1T, D, Hd, w.dh2, D, R, D, Hd, nullptr, 0, st));                    // dh2 = dpre W1
  else EP_TRY(gemm(true, false, vg(dpre, Hd, pr.fc1_w, D, w.dh2, D, R, D, Hd), 1, st));
  EP_TRY(dv_colsum(w.dh2, w.x1, w.stat2, R, D, acc, gr.n2_w, gr.n2_b, w.scr, st));
  // d x1 = g0 (broadcast) + LayerNorm backward of dh2
  hipLaunchKernelGGL(ep_dv_bcast_kernel, dim3((unsigned)((nd / 4 + 255) / 256)), dim3(256), 0, st, w.g0, nd / 4, N, D / 4, 1.0f, w.dx1);
  hipLaunchKernelGGL(ep_rowln_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, st, w.dh2, w.x1, w.stat2, pr.n2_w, w.dx1, R, D, w.dx1);
  EP_LAUNCH_CHECK("ep_dinovit LN2 backward kernels");
  // x1 = x + ctx Wp^T + bp
  EP_TRY(dv_colsum(w.dx1, nullptr, nullptr, R, D, acc, gr.proj_b, nullptr, w.scr, st));
  { GemmParams g = vg(w.dx1, D, w.ctx, D, gr.proj_w, D, D, D, R); g.accumulate = acc; g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(false, false, g, 1, st)); }                                                               // d Wp
  if (pl) EP_TRY(dv_pl(w.dx1, D, w.plPT, D, D, w.dctx, D, R, D, D, nullptr, 0, st));                      // dctx = dx1 Wp
  else EP_TRY(gemm(true, false, vg(w.dx1, D, pr.proj_w, D, w.dctx, D, R, D, D), 1, st));
  const int64_t s3 = (int64_t)N * 3 * D, sS = (int64_t)H * N * N, sD = (int64_t)N * D;
  for (int hh = 0; hh < H; ++hh) {
    { GemmParams g = vg(w.dctx + hh * dh, D, w.QKV + 2 * D + hh * dh, 3 * D, w.dS + (int64_t)hh * N * N, N, N, N, dh);           // dA_h = dctx_h v_h^T
      g.sAz = sD; g.sBz = s3; g.sCz = sS; EP_TRY(gemm(true, true, g, B, st)); }
    { GemmParams g = vg(w.S + (int64_t)hh * N * N, N, w.dctx + hh * dh, D, w.dQKV + 2 * D + hh * dh, 3 * D, N, dh, N);            // dv_h = A_h^T dctx_h
      g.sAz = sS; g.extA = N; g.sBz = sD; g.extB = dh; g.sCz = s3; EP_TRY(gemm(false, false, g, B, st)); }
  }
  hipLaunchKernelGGL(ep_dv_softmax_bwd_kernel, dim3((unsigned)(((int64_t)R * H + 3) / 4)), dim3(256), 0, st, w.S, w.dS, (int64_t)R * H, N);
  EP_LAUNCH_CHECK("ep_dv_softmax_bwd_kernel");
  for (int hh = 0; hh < H; ++hh) {
    { GemmParams g = vg(w.dS + (int64_t)hh * N * N, N, w.QKV + D + hh * dh, 3 * D, w.dQKV + hh * dh, 3 * D, N, dh, N);            // dq_h = scale dS_h k_h
      g.sAz = sS; g.sBz = s3; g.sCz = s3; g.extB = dh; g.alpha = scale; EP_TRY(gemm(true, false, g, B, st)); }
    { GemmParams g = vg(w.dS + (int64_t)hh * N * N, N, w.QKV + hh * dh, 3 * D, w.dQKV + D + hh * dh, 3 * D, N, dh, N);            // dk_h = scale dS_h^T q_h
      g.sAz = sS; g.extA = N; g.sBz = s3; g.extB = dh; g.sCz = s3; g.alpha = scale; EP_TRY(gemm(false, false, g, B, st)); }
  }
  // QKV = h Wqkv^T ; h = LayerNorm1(x): the tokens are frozen, only the affine parameters take a gradient
  { GemmParams g = vg(w.dQKV, 3 * D, w.h, D, gr.qkv_w, D, 3 * D, D, R); g.accumulate = acc; g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(false, false, g, 1, st)); }
  if (pl) EP_TRY(dv_pl(w.dQKV, 3 * D, w.plQT, D, 3 * D, w.dh2, D, R, D, 3 * D, nullptr, 0, st));          // dh (reusing dh2)
  else EP_TRY(gemm(true, false, vg(w.dQKV, 3 * D, pr.qkv_w, D, w.dh2, D, R, D, 3 * D), 1, st));
  EP_TRY(dv_colsum(w.dh2, x, w.stat1, R, D, acc, gr.n1_w, gr.n1_b, w.scr, st));
  return 0;
}

static ep_dinovit_params dv_views(float* base, const int64_t o[DV_NT]) {
  ep_dinovit_params p;
  p.n1_w = base + o[0]; p.n1_b = base + o[1]; p.qkv_w = base + o[2]; p.proj_w = base + o[3]; p.proj_b = base + o[4];
  p.n2_w = base + o[5]; p.n2_b = base + o[6]; p.fc1_w = base + o[7]; p.fc1_b = base + o[8]; p.fc2_w = base + o[9];
  p.fc2_b = base + o[10];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

static int dv_tokens_ok(const ep_dinovit_dims& d, const void* x, int x_dtype, int64_t x_bstride) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, 1));
  EP_REQUIRE(x_dtype == EP_DTYPE_F32 && x_bstride == (int64_t)d.N * d.D, EP_E_UNSUPPORTED,
             "dinovit: the tokens feed matrix-core contractions: a dense fp32 (B, N, D) tensor is required");
  return 0;
}

size_t ep_dinovit_pool_workspace_bytes(const ep_dinovit_dims* dims) {
  if (!dims || dv_check(*dims, false) != 0) return 0;
  return dv_carve(*dims, nullptr, false).total;
}

int ep_dinovit_pool_forward(const ep_dinovit_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const ep_dinovit_params* params, float* out, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && out && ws, EP_E_ARG, "ep_dinovit_pool_forward: null pointer");
  EP_TRY(dv_check(*dims, false));
  EP_TRY(dv_params_ok(params, "ep_dinovit_pool_forward"));
  EP_TRY(dv_tokens_ok(*dims, x, x_dtype, x_bstride));
  EP_REQUIRE(aligned16(ws) && aligned16(out), EP_E_ALIGN, "ep_dinovit_pool_forward: out / ws must be 16-byte aligned");
  const DvWs w = dv_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dinovit_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return dv_forward_core(*dims, static_cast<const float*>(x), *params, w, out, (hipStream_t)stream);
}

int ep_dinovit_pool_backward(const ep_dinovit_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                             const ep_dinovit_params* params, const float* dout, const ep_dinovit_params* grads, int accumulate,
                             void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dout && ws, EP_E_ARG, "ep_dinovit_pool_backward: null pointer");
  EP_TRY(dv_check(*dims, false));
  EP_TRY(dv_params_ok(params, "ep_dinovit_pool_backward(params)"));
  EP_TRY(dv_params_ok(grads, "ep_dinovit_pool_backward(grads)"));
  EP_TRY(dv_tokens_ok(*dims, x, x_dtype, x_bstride));
  EP_REQUIRE(aligned16(ws) && aligned16(dout), EP_E_ALIGN, "ep_dinovit_pool_backward: dout / ws must be 16-byte aligned");
  const DvWs w = dv_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dinovit_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return dv_backward_core(*dims, static_cast<const float*>(x), *params, dout, *grads, accumulate, w, (hipStream_t)stream);
}

/* attention weights (B, H, N, N) of the last forward on this workspace (block.py return_attention) */
int ep_dinovit_attention(const ep_dinovit_dims* dims, const void* ws, float* A, ep_stream_t stream) {
  EP_REQUIRE(dims && ws && A, EP_E_ARG, "ep_dinovit_attention: null pointer");
  EP_TRY(dv_check(*dims, false));
  const DvWs w = dv_carve(*dims, const_cast<void*>(ws), false);
  EP_HIP(hipMemcpyAsync(A, w.S, (size_t)dims->B * dims->H * dims->N * dims->N * sizeof(float), hipMemcpyDeviceToDevice,
                        (hipStream_t)stream));
  return 0;
}

int64_t ep_dinovit_head_param_offsets(const ep_dinovit_dims* dims, int64_t offsets[13]) { return dv_offsets(*dims, offsets); }

size_t ep_dinovit_head_workspace_bytes(const ep_dinovit_dims* dims) {
  if (!dims || dv_check(*dims, true) != 0) return 0;
  return dv_carve(*dims, nullptr, true).total;
}

int ep_dinovit_head_train_step(const ep_dinovit_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_dinovit_head_train_step: null pointer");
  const ep_dinovit_dims& d = s->dims;
  EP_TRY(dv_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const DvWs w = dv_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dinovit_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[DV_NT];
  const int64_t total = dv_offsets(d, offs);
  const ep_dinovit_params pr = dv_views(s->params, offs), gr = dv_views(s->grads, offs);
  float* Wc = s->params + offs[11]; float* bc = s->params + offs[12];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(dv_tokens_ok(d, s->x, s->x_dtype, s->x_bstride));
    const float* x = static_cast<const float*>(s->x);
    EP_TRY(dv_forward_core(d, x, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, s->grads + offs[11], s->grads + offs[12],
                           s->accumulate, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    EP_TRY(dv_backward_core(d, x, pr, w.dy, gr, s->accumulate, w, st));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    int64_t sizes[DV_NT];
    dv_sizes(d, sizes);
    const int trust[DV_NT] = {0, 0, 1, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0};       // util/lars.py:22: ndim > 1
    ep_segment segs[DV_NT];
    for (int i = 0; i < DV_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? DV_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_dinovit_head_eval_forward(const ep_dinovit_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const float* params,
                                 const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                                 void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_dinovit_head_eval_forward: null pointer");
  const ep_dinovit_dims& d = *dims;
  EP_TRY(dv_check(d, true));
  EP_TRY(dv_tokens_ok(d, x, x_dtype, x_bstride));
  const DvWs w = dv_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dinovit_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[DV_NT];
  dv_offsets(d, offs);
  const ep_dinovit_params pr = dv_views(const_cast<float*>(params), offs);
  EP_TRY(dv_forward_core(d, static_cast<const float*>(x), pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[11], params + offs[12], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
