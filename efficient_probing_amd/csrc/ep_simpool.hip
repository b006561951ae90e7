// SimPool heads (reference poolings/simpool.py; registry entries probe_heads.py:66-70):
//   simpool  = SimPool(dim, num_heads=1, qkv_bias=False, gamma=None)            (simpool.py:5-91)
//   esimpool = SimPool_nolinears(dim, num_heads=12, gamma=None)                 (simpool.py:93-170)
// Both take the mean token gap[b] = mean_n x[b,n] as the query source and LayerNorm (eps 1e-6) the patch tokens.
//
// simpool (one head):   q = Wq gap ; k = Wk LN(x) ; v = LN(x) ; out = softmax(scale q.k) v
//     score[b,n] = scale (Wq gap[b]) . (Wk (g * xhat[b,n] + beta)) = u[b] . xhat[b,n] + const,
//     u[b] = scale g * t[b],  t[b] = Wk^T (Wq gap[b])          (two (B x D x D) contractions)
//     out[b] = g * Phat[b] + beta,   Phat[b] = sum_n A[b,n] xhat[b,n]
// esimpool (H heads, no linear maps):  q = LN(gap) ; k = LN(x) ; v = x, heads are channel slices
//     score[b,h,n] = scale sum_{c in h} q[b,c] (g_c xhat[b,n,c] + beta_c) = u[b, slice h] . xhat[b,n, slice h] + const,
//     u[b,c] = scale q[b,c] g_c,   out[b, slice h] = sum_n A[b,h,n] x[b,n, slice h]            (raw tokens pooled)
// The query rows differ per image and the backward needs d u per image: the per-image-query token passes
// (ep_pool_imgq.hip) with LayerNorm-of-tokens scores.  Everything the passes need from the tokens besides the tokens
// -- per-token {mean, rstd} and the per-image mean token -- depends on the frozen tokens only, so a resident token
// store computes both ONCE (ep_token_stats, ep_channel_stats) and a step is exactly two streaming reads.
#include "ep_side.h"
#include "ep_pool_imgq.h"

namespace ep {

int token_image_stats(const void* x, int x_dtype, int64_t bstride, int B, int N, int D, float eps, int mode, float* stats, float* out,
                      hipStream_t st);   // ep_aim.hip
int channel_stats(const void* x, int x_dtype, int64_t bstride, const int32_t* index, int B, int N, int D, float* img,
                  hipStream_t st);                                   // ep_aim.hip

// out[b, :] = table[index[b], 0:D]   (rows of a cached per-image table, row stride ld)
__global__ void ep_esp_gather_kernel(const float* __restrict__ table, int64_t ld, const int* __restrict__ index, int64_t total,
                                     int D, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = table[(int64_t)index[i / D] * ld + (i % D)];
}

// u = scale g * t   (element-wise over (B, D))
__global__ void ep_sp_u_kernel(const float* __restrict__ t, const float* __restrict__ g, int64_t total, int D, float scale,
                               float* __restrict__ u) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) u[i] = scale * g[i % D] * t[i];
}
// y = g * Phat + beta
__global__ void ep_sp_out_kernel(const float* __restrict__ Ph, const float* __restrict__ g, const float* __restrict__ beta,
                                 int64_t total, int D, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) y[i] = fmaf(g[i % D], Ph[i], beta[i % D]);
}
// dPhat = dy * g
__global__ void ep_sp_dphat_kernel(const float* __restrict__ dy, const float* __restrict__ g, int64_t total, int D,
                                   float* __restrict__ dPh) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dPh[i] = dy[i] * g[i % D];
}
// simpool, after the second pass:  dt = scale g * du ;  d g (+)= sum_b (dy Phat + scale du t) ;  d beta (+)= sum_b dy
__global__ __launch_bounds__(1024) void ep_sp_colgrad_kernel(const float* __restrict__ dy, const float* __restrict__ Ph,
                                                           const float* __restrict__ du, const float* __restrict__ t,
                                                           const float* __restrict__ g, int B, int D, float scale,
                                                           int accumulate, float* __restrict__ dt, float* __restrict__ dg,
                                                           float* __restrict__ dbeta) {
  __shared__ float sm[32][33];                       // 32 column lanes x 32 row lanes (grid (D + 31) / 32)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  const bool ok = c < D;
  float sg = 0.f, sb = 0.f;
  if (ok) {
    const float gc = g[c] * scale;
    for (int b = ty; b < B; b += 32) {
      const int64_t i = (int64_t)b * D + c;
      const float d_ = du[i], y_ = dy[i];
      dt[i] = gc * d_;
      sg += fmaf(y_, Ph[i], scale * d_ * t[i]);
      sb += y_;
    }
  }
  __syncthreads(); sm[ty][tx] = sg; __syncthreads();
  float tg = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) tg += sm[i][tx];
  __syncthreads(); sm[ty][tx] = sb; __syncthreads();
  float tb = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) tb += sm[i][tx];
  if (ty == 0 && ok) {
    dg[c] = accumulate ? dg[c] + tg : tg;
    dbeta[c] = accumulate ? dbeta[c] + tb : tb;
  }
}

// esimpool: ghat[b] = (gap[b] - mean) rstd over the D channels (LayerNorm of the mean token, biased variance, eps inside
// the root), q = g * ghat + beta, u = scale q g.   One wave per image.
__global__ __launch_bounds__(256) void ep_esp_q_kernel(const float* __restrict__ gap, int64_t gap_ld, const int* __restrict__ index,
                                                     const float* __restrict__ g, const float* __restrict__ beta, int B, int D,
                                                     float eps, float scale, float* __restrict__ ghat, float* __restrict__ q,
                                                     float* __restrict__ u) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const float* row = gap + (int64_t)(index ? index[b] : b) * gap_ld;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += row[d];
  const float mean = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int d = lane; d < D; d += 64) { const float e = row[d] - mean; v = fmaf(e, e, v); }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
  for (int d = lane; d < D; d += 64) {
    const float gh = (row[d] - mean) * rstd;
    const float qq = fmaf(g[d], gh, beta[d]);
    ghat[(int64_t)b * D + d] = gh;
    q[(int64_t)b * D + d] = qq;
    u[(int64_t)b * D + d] = scale * qq * g[d];
  }
}
// esimpool, after the second pass:  d g_c (+)= scale sum_b du (q + g ghat) ;  d beta_c (+)= scale g_c sum_b du
__global__ __launch_bounds__(256) void ep_esp_colgrad_kernel(const float* __restrict__ du, const float* __restrict__ q,
                                                           const float* __restrict__ ghat, const float* __restrict__ g, int B,
                                                           int D, float scale, int accumulate, float* __restrict__ dg,
                                                           float* __restrict__ dbeta) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * CG + tx;
  const bool ok = c < D;
  float sg = 0.f, sb = 0.f;
  if (ok) {
    const float gc = g[c];
    for (int b = ty; b < B; b += RL) {
      const int64_t i = (int64_t)b * D + c;
      const float d_ = du[i];
      sg += d_ * fmaf(gc, ghat[i], q[i]);
      sb += d_;
    }
  }
  sg = colreduce(sg, sm, tx, ty);
  sb = colreduce(sb, sm, tx, ty);
  if (ty == 0 && ok) {
    const float vg = scale * sg, vb = scale * g[c] * sb;
    dg[c] = accumulate ? dg[c] + vg : vg;
    dbeta[c] = accumulate ? dbeta[c] + vb : vb;
  }
}

// attention weights A[b,h,n] of the last forward, recomputed from u and the saved softmax state (one wave per (b,h,n) row
// would be wasteful: one workgroup per (image, head), threads over tokens; the slice dot product is short)
__global__ __launch_bounds__(256) void ep_sp_attention_kernel(ImgqParams p, float* __restrict__ A) {
  const int b = blockIdx.x, h = blockIdx.y;
  const int D = p.D, N = p.N, H = p.H, Dh = D / H;
  const int64_t img = (int64_t)(p.index ? p.index[b] : b);
  const float* ts = p.tokstat ? p.tokstat + img * N * 2 : nullptr;
  const float* u = p.u + (int64_t)b * D + h * Dh;
  const float m = p.ML[((int64_t)b * H + h) * 2], inv = 1.0f / p.ML[((int64_t)b * H + h) * 2 + 1];
  float usum = 0.f;
  for (int c = 0; c < Dh; ++c) usum += u[c];
  for (int n = threadIdx.x; n < N; n += 256) {
    float d = 0.f;
    if (p.x_bf16) {
      const uint16_t* row = static_cast<const uint16_t*>(p.x) + img * p.x_bstride + (int64_t)n * D + h * Dh;
      for (int c = 0; c < Dh; ++c) d = fmaf(u[c], __uint_as_float((unsigned)row[c] << 16), d);
    } else {
      const float* row = static_cast<const float*>(p.x) + img * p.x_bstride + (int64_t)n * D + h * Dh;
      for (int c = 0; c < Dh; ++c) d = fmaf(u[c], row[c], d);
    }
    const float s = ts ? ts[2 * n + 1] * (d - ts[2 * n] * usum) : d;
    A[((int64_t)b * H + h) * N + n] = __builtin_amdgcn_exp2f((s - m) * 1.4426950408889634f) * inv;
  }
}

// ---------------------------------------------------------------------------------------------
constexpr int SP_NT = 6;                             // norm.weight | norm.bias | wq.weight | wk.weight | fc.weight | fc.bias
struct SpWs {
  float *img, *stats, *qq, *t, *u, *Ph, *ML, *dPh, *du, *dt, *dqq, *ghat;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static int64_t sp_offsets(const ep_simpool_dims& d, int64_t offs[SP_NT]) {
  const int64_t D = d.D, DD = d.linears ? D * D : 0;
  const int64_t sizes[SP_NT] = {D, D, DD, DD, (int64_t)d.C * D, d.C};
  int64_t off = 0;
  for (int i = 0; i < SP_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static SpWs sp_carve(const ep_simpool_dims& d, void* base, bool head) {
  SpWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D;
  w.img = take(B * 2 * D); w.stats = take(B * d.N * 2);
  w.qq = take(B * D); w.t = take(B * D); w.u = take(B * D); w.Ph = take(B * D); w.ML = take(B * d.H * 2);
  w.dPh = take(B * D); w.du = take(B * D); w.dt = take(B * D); w.dqq = take(B * D); w.ghat = take(B * D);
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[SP_NT];
    w.opt_ws_bytes = optim_workspace_bytes(sp_offsets(d, offs), SP_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int sp_check(const ep_simpool_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0, EP_E_ARG, "simpool dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && d.D % 4 == 0, EP_E_SHAPE, "simpool: D %% H != 0 (D=%d H=%d) -- the reference's reshape fails too", d.D, d.H);
  EP_REQUIRE(!d.linears || d.H == 1, EP_E_UNSUPPORTED, "simpool with linear maps: one head (what the registry builds)");
  EP_REQUIRE(imgq_supported(d.D, d.H), EP_E_UNSUPPORTED, "simpool: no token-pass kernel for D=%d with %d heads", d.D, d.H);
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "simpool head: C must be positive");
  return 0;
}

static int sp_params_ok(const ep_simpool_dims& d, const ep_simpool_params* p, const char* what) {
  EP_REQUIRE(p && p->norm_w && p->norm_b && (!d.linears || (p->wq && p->wk)), EP_E_ARG, "%s: null tensor", what);
  EP_REQUIRE(aligned16(p->norm_w) && aligned16(p->norm_b) && aligned16(p->wq) && aligned16(p->wk), EP_E_ALIGN,
             "%s: tensors must be 16-byte aligned", what);
  return 0;
}

static GemmParams sg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

struct SpCache { const float* token_stats; const float* image_stats; };   // optional per-store tables (indexed like x)

static ImgqParams sp_pass(const ep_simpool_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                          const float* tokstat, const SpWs& w) {
  ImgqParams q{};
  q.x = x; q.x_bf16 = x_dtype == EP_DTYPE_BF16 ? 1 : 0; q.x_bstride = bstride; q.index = index;
  q.B = d.B; q.N = d.N; q.D = d.D; q.H = d.H;
  q.u = w.u; q.tokstat = tokstat; q.pool_ln = d.linears ? 1 : 0; q.P = w.Ph; q.ML = w.ML;
  return q;
}

// resolves the two per-token / per-image tables: the caller's cached ones or computed here for the batch
static int sp_tables(const ep_simpool_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                     const SpCache& c, float ln_eps, const SpWs& w, bool compute, const float*& tokstat, const float*& gap,
                     int64_t& gap_ld, const int32_t*& gap_index, hipStream_t st) {
  tokstat = c.token_stats;
  if (!tokstat && !c.image_stats && !index && (d.D + 255) / 256 <= 5) {
    // neither table is cached: both in ONE read of the batch (same bits as the two separate kernels)
    if (compute) EP_TRY(token_image_stats(x, x_dtype, bstride, d.B, d.N, d.D, ln_eps, 0, w.stats, w.img, st));
    tokstat = w.stats; gap = w.img; gap_index = nullptr; gap_ld = 2 * (int64_t)d.D;
    return 0;
  }
  if (!tokstat) {
    EP_REQUIRE(!index, EP_E_ARG, "simpool: an indexed token store needs precomputed token statistics");
    if (compute) EP_TRY(token_stats(x, x_dtype == EP_DTYPE_BF16, bstride, d.B, d.N, d.D, ln_eps, w.stats, st));
    tokstat = w.stats;
  }
  gap_ld = 2 * (int64_t)d.D;
  if (c.image_stats) { gap = c.image_stats; gap_index = index; }
  else {
    if (compute) EP_TRY(channel_stats(x, x_dtype, bstride, index, d.B, d.N, d.D, w.img, st));
    gap = w.img; gap_index = nullptr;
  }
  return 0;
}

static int sp_forward_core(const ep_simpool_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                           const SpCache& c, float ln_eps, const ep_simpool_params& pr, const SpWs& w, float* y,
                           hipStream_t st) {
  const int D = d.D, B = d.B;
  const float scale = (float)pow((double)(D / d.H), -0.5);                 // simpool.py:9-10 / :97-98
  const float* tokstat; const float* gap; int64_t gap_ld; const int32_t* gidx;
  EP_TRY(sp_tables(d, x, x_dtype, bstride, index, c, ln_eps, w, true, tokstat, gap, gap_ld, gidx, st));
  const int64_t total = (int64_t)B * D;
  const unsigned eg = (unsigned)((total + 255) / 256);
  if (d.linears) {
    if (gidx) {                                       // gather the cached mean tokens of the batch (B x D, tiny)
      hipLaunchKernelGGL(ep_esp_gather_kernel, dim3(eg), dim3(256), 0, st, gap, gap_ld, gidx, total, D, w.ghat);
      gap = w.ghat; gap_ld = D;
    }
    EP_TRY(gemm(true, true, sg(gap, gap_ld, pr.wq, D, w.qq, D, B, D, D), 1, st));            // qq = gap Wq^T
    { GemmParams g = sg(w.qq, D, pr.wk, D, w.t, D, B, D, D); g.extB = D; EP_TRY(gemm(true, false, g, 1, st)); }   // t = qq Wk
    hipLaunchKernelGGL(ep_sp_u_kernel, dim3(eg), dim3(256), 0, st, w.t, pr.norm_w, total, D, scale, w.u);
  } else {
    hipLaunchKernelGGL(ep_esp_q_kernel, dim3((B + 3) / 4), dim3(256), 0, st, gap, gap_ld, gidx, pr.norm_w, pr.norm_b, B, D,
                       ln_eps, scale, w.ghat, w.qq, w.u);
  }
  EP_LAUNCH_CHECK("simpool query kernels");
  ImgqParams q = sp_pass(d, x, x_dtype, bstride, index, tokstat, w);
  if (!d.linears) q.P = y;                            // raw pooled slices ARE the output
  EP_TRY(imgq_forward(q, st));
  if (d.linears) {
    hipLaunchKernelGGL(ep_sp_out_kernel, dim3(eg), dim3(256), 0, st, w.Ph, pr.norm_w, pr.norm_b, total, D, y);
    EP_LAUNCH_CHECK("ep_sp_out_kernel");
  }
  return 0;
}

// `y`: the forward's output (esimpool: the pooled slices the softmax correction needs)
static int sp_backward_core(const ep_simpool_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                            const SpCache& c, float ln_eps, const ep_simpool_params& pr, const float* y, const float* dy,
                            const ep_simpool_params& gr, int acc, const SpWs& w, hipStream_t st) {
  const int D = d.D, B = d.B;
  const float scale = (float)pow((double)(D / d.H), -0.5);
  const float* tokstat; const float* gap; int64_t gap_ld; const int32_t* gidx;
  EP_TRY(sp_tables(d, x, x_dtype, bstride, index, c, ln_eps, w, false, tokstat, gap, gap_ld, gidx, st));   // left by the forward
  const int64_t total = (int64_t)B * D;
  const unsigned eg = (unsigned)((total + 255) / 256);
  ImgqParams q = sp_pass(d, x, x_dtype, bstride, index, tokstat, w);
  q.du = w.du;
  if (d.linears) {
    hipLaunchKernelGGL(ep_sp_dphat_kernel, dim3(eg), dim3(256), 0, st, dy, pr.norm_w, total, D, w.dPh);
    q.dP = w.dPh;
    EP_TRY(imgq_backward(q, st));
    hipLaunchKernelGGL(ep_sp_colgrad_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, dy, w.Ph, w.du, w.t, pr.norm_w, B, D,
                       scale, acc, w.dt, gr.norm_w, gr.norm_b);
    EP_LAUNCH_CHECK("simpool backward kernels");
    if (gidx) { gap = w.ghat; gap_ld = D; }           // gathered by the forward
    { GemmParams g = sg(w.qq, D, w.dt, D, gr.wk, D, D, D, B); g.accumulate = acc; EP_TRY(gemm(false, false, g, 1, st)); }    // dWk = qq^T dt
    EP_TRY(gemm(true, true, sg(w.dt, D, pr.wk, D, w.dqq, D, B, D, D), 1, st));                                             // dqq = dt Wk^T
    { GemmParams g = sg(w.dqq, D, gap, gap_ld, gr.wq, D, D, D, B); g.extB = D; g.accumulate = acc; EP_TRY(gemm(false, false, g, 1, st)); }   // dWq = dqq^T gap
  } else {
    q.P = const_cast<float*>(y); q.dP = dy;
    EP_TRY(imgq_backward(q, st));
    hipLaunchKernelGGL(ep_esp_colgrad_kernel, dim3((D + CG - 1) / CG), dim3(256), 0, st, w.du, w.qq, w.ghat, pr.norm_w, B, D,
                       scale, acc, gr.norm_w, gr.norm_b);
    EP_LAUNCH_CHECK("esimpool backward kernels");
  }
  return 0;
}

static ep_simpool_params sp_views(const ep_simpool_dims& d, float* base, const int64_t o[SP_NT]) {
  ep_simpool_params p;
  p.norm_w = base + o[0]; p.norm_b = base + o[1];
  p.wq = d.linears ? base + o[2] : nullptr; p.wk = d.linears ? base + o[3] : nullptr;
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

static int imgq_args(const void* x, int x_dtype, int64_t x_bstride, int B, int N, int D, int H, const float* u,
                     const float* token_stats_, int pool_ln, float* P, float* ML) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, 1));
  EP_REQUIRE(u && P && ML && H > 0, EP_E_ARG, "per-image-query token pass: null pointer");
  EP_REQUIRE(aligned16(u) && aligned16(P), EP_E_ALIGN, "per-image-query token pass: u / P must be 16-byte aligned");
  EP_REQUIRE(D % H == 0 && imgq_supported(D, H), EP_E_UNSUPPORTED, "per-image-query token pass: D=%d with %d heads is not supported", D, H);
  EP_REQUIRE(!pool_ln || token_stats_, EP_E_ARG, "per-image-query token pass: pooling normalised tokens needs token statistics");
  return 0;
}

int ep_imgq_pool_forward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D, int H,
                         const float* u, const float* token_stats_, int pool_ln, float* P, float* ML, ep_stream_t stream) {
  EP_TRY(imgq_args(x, x_dtype, x_bstride, B, N, D, H, u, token_stats_, pool_ln, P, ML));
  ImgqParams q{};
  q.x = x; q.x_bf16 = x_dtype == EP_DTYPE_BF16 ? 1 : 0; q.x_bstride = x_bstride; q.index = image_index;
  q.B = B; q.N = N; q.D = D; q.H = H; q.u = u; q.tokstat = token_stats_; q.pool_ln = pool_ln; q.P = P; q.ML = ML;
  return imgq_forward(q, (hipStream_t)stream);
}

int ep_imgq_pool_backward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D, int H,
                          const float* u, const float* token_stats_, int pool_ln, const float* P, const float* ML,
                          const float* dP, float* du, ep_stream_t stream) {
  EP_TRY(imgq_args(x, x_dtype, x_bstride, B, N, D, H, u, token_stats_, pool_ln, const_cast<float*>(P), const_cast<float*>(ML)));
  EP_REQUIRE(dP && du && aligned16(dP) && aligned16(du), EP_E_ARG, "ep_imgq_pool_backward: dP / du null or not 16-byte aligned");
  ImgqParams q{};
  q.x = x; q.x_bf16 = x_dtype == EP_DTYPE_BF16 ? 1 : 0; q.x_bstride = x_bstride; q.index = image_index;
  q.B = B; q.N = N; q.D = D; q.H = H; q.u = u; q.tokstat = token_stats_; q.pool_ln = pool_ln;
  q.P = const_cast<float*>(P); q.ML = const_cast<float*>(ML); q.dP = dP; q.du = du;
  return imgq_backward(q, (hipStream_t)stream);
}

static int rowq_params(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D, int Q,
                       const float* tokstat, PoolParams& p) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, Q));
  p = pool_params(x, x_bstride, B, N, D, Q, 1.0f, x_dtype);
  p.index = image_index; p.tokstat = tokstat; p.cls_bstride = (int64_t)Q * D;
  EP_REQUIRE(imgqf_supported(p), EP_E_UNSUPPORTED, "full-width per-image-query pass: 1 <= Q <= 4, D %% 4 == 0, D <= 1280 (Q=%d D=%d)",
             Q, D);
  return 0;
}

int ep_rowq_pool_forward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D, int Q,
                         const float* u, const float* token_stats_, const float* score_bias, float* P, float* S, float* ML,
                         ep_stream_t stream) {
  EP_REQUIRE(u && P && S && ML && aligned16(u) && aligned16(P) && aligned16(ML), EP_E_ARG,
             "ep_rowq_pool_forward: u / P / S / ML null or not 16-byte aligned");
  PoolParams p;
  EP_TRY(rowq_params(x, x_dtype, x_bstride, image_index, B, N, D, Q, token_stats_, p));
  p.cls = u; p.sbias = score_bias; p.P = P; p.S = S; p.ML = ML;
  return imgqf_forward(p, (hipStream_t)stream);
}

int ep_rowq_pool_backward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D, int Q,
                          const float* token_stats_, const float* S, const float* ML, const float* dP, const float* dA_bias,
                          float* dS_out, float* du, ep_stream_t stream) {
  EP_REQUIRE(S && ML && dP && du && aligned16(dP) && aligned16(du) && aligned16(ML), EP_E_ARG,
             "ep_rowq_pool_backward: S / ML / dP / du null or not 16-byte aligned");
  PoolParams p;
  EP_TRY(rowq_params(x, x_dtype, x_bstride, image_index, B, N, D, Q, token_stats_, p));
  p.S = const_cast<float*>(S); p.ML = const_cast<float*>(ML); p.dP = dP; p.dabias = dA_bias; p.dSout = dS_out;
  return imgqf_backward(p, du, (hipStream_t)stream);
}

size_t ep_simpool_pool_workspace_bytes(const ep_simpool_dims* dims) {
  if (!dims || sp_check(*dims, false) != 0) return 0;
  return sp_carve(*dims, nullptr, false).total;
}

int ep_simpool_pool_forward(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const int32_t* image_index, const float* token_stats_, const float* image_stats, float ln_eps,
                            const ep_simpool_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_simpool_pool_forward: null pointer");
  EP_TRY(sp_check(*dims, false));
  EP_TRY(sp_params_ok(*dims, params, "ep_simpool_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, 1));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_simpool_pool_forward: y / ws must be 16-byte aligned");
  const SpWs w = sp_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_simpool_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return sp_forward_core(*dims, x, x_dtype, x_bstride, image_index, SpCache{token_stats_, image_stats}, ln_eps, *params, w, y,
                         (hipStream_t)stream);
}

int ep_simpool_pool_backward(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                             const int32_t* image_index, const float* token_stats_, const float* image_stats, float ln_eps,
                             const ep_simpool_params* params, const float* y, const float* dy,
                             const ep_simpool_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && dy && ws, EP_E_ARG, "ep_simpool_pool_backward: null pointer");
  EP_TRY(sp_check(*dims, false));
  EP_TRY(sp_params_ok(*dims, params, "ep_simpool_pool_backward(params)"));
  EP_TRY(sp_params_ok(*dims, grads, "ep_simpool_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, 1));
  EP_REQUIRE(aligned16(ws) && aligned16(dy) && aligned16(y), EP_E_ALIGN, "ep_simpool_pool_backward: y / dy / ws must be 16-byte aligned");
  const SpWs w = sp_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_simpool_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return sp_backward_core(*dims, x, x_dtype, x_bstride, image_index, SpCache{token_stats_, image_stats}, ln_eps, *params, y, dy,
                          *grads, accumulate, w, (hipStream_t)stream);
}

int ep_simpool_attention(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* token_stats_, const void* ws, float* A,
                         ep_stream_t stream) {
  EP_REQUIRE(dims && x && ws && A, EP_E_ARG, "ep_simpool_attention: null pointer");
  EP_TRY(sp_check(*dims, false));
  const SpWs w = sp_carve(*dims, const_cast<void*>(ws), false);
  ImgqParams q = sp_pass(*dims, x, x_dtype, x_bstride, image_index, token_stats_ ? token_stats_ : w.stats, w);
  hipLaunchKernelGGL(ep_sp_attention_kernel, dim3(dims->B, dims->H), dim3(256), 0, (hipStream_t)stream, q, A);
  EP_LAUNCH_CHECK("ep_sp_attention_kernel");
  return 0;
}

int64_t ep_simpool_head_param_offsets(const ep_simpool_dims* dims, int64_t offsets[6]) { return sp_offsets(*dims, offsets); }

size_t ep_simpool_head_workspace_bytes(const ep_simpool_dims* dims) {
  if (!dims || sp_check(*dims, true) != 0) return 0;
  return sp_carve(*dims, nullptr, true).total;
}

int ep_simpool_head_train_step(const ep_simpool_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_simpool_head_train_step: null pointer");
  const ep_simpool_dims& d = s->dims;
  EP_TRY(sp_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const SpWs w = sp_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_simpool_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[SP_NT];
  const int64_t total = sp_offsets(d, offs);
  const ep_simpool_params pr = sp_views(d, s->params, offs), gr = sp_views(d, s->grads, offs);
  float* Wc = s->params + offs[4]; float* bc = s->params + offs[5];
  const SpCache cache{s->token_stats, s->image_stats};
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, 1));
    EP_TRY(sp_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, cache, s->ln_eps, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, s->grads + offs[4], s->grads + offs[5],
                           s->accumulate, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    EP_TRY(sp_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, cache, s->ln_eps, pr, w.y, w.dy, gr,
                            s->accumulate, w, st));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t D = d.D, DD = d.linears ? D * D : 0;
    const int64_t sizes[SP_NT] = {D, D, DD, DD, (int64_t)d.C * D, d.C};
    const int trust[SP_NT] = {0, 0, 1, 1, 1, 0};     // util/lars.py:22: trust ratio + weight decay for ndim > 1
    ep_segment segs[SP_NT];
    int n = 0;
    for (int i = 0; i < SP_NT; ++i) if (sizes[i] > 0) segs[n++] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? n : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_simpool_head_eval_forward(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                                 const int32_t* image_index, const float* token_stats_, const float* image_stats,
                                 float ln_eps, const float* params, const float* running_mean, const float* running_var,
                                 float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_simpool_head_eval_forward: null pointer");
  const ep_simpool_dims& d = *dims;
  EP_TRY(sp_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, 1));
  const SpWs w = sp_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_simpool_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[SP_NT];
  sp_offsets(d, offs);
  const ep_simpool_params pr = sp_views(d, const_cast<float*>(params), offs);
  EP_TRY(sp_forward_core(d, x, x_dtype, x_bstride, image_index, SpCache{token_stats_, image_stats}, ln_eps, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[4], params + offs[5], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
