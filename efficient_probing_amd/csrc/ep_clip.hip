// CLIP attention pooling (reference poolings/clip/attention_pool2d.py:100-169 AttentionPool2d; registry entry
// probe_heads.py:54-57,71: AttentionPool2d(in_features=dim, feat_size=14 | 16) -> 4 heads, qkv bias, learned absolute
// position embedding, LayerNorm eps 1e-6 in front).  With xn = LayerNorm(x) (affine g, b), the N + 1 rows
//     t_0 = mean_n xn + pos_0 ,   t_n = xn_n + pos_n
// go through one qkv Linear and ONLY row 0 of the attention output is returned (attention_pool2d.py:169): the query is the
// image's own mean row, so the query rows differ per image (as in SimPool) but span the full width (4 heads x D):
//     w[b,h]   = scale Wk_h^T (Wq t_0[b] + bq)_h                                   (B, H, D)
//     s[b,h,n] = w[b,h] . t_n + const = (g * w[b,h]) . xhat[b,n] + w[b,h] . pos_n   (+ a constant that cancels)
//     o[b,h]   = Wv_h (g * (sum_n A_n xhat_n) + sum_n A_n pos_n + b) + bv_h         (sum over the N + 1 entries)
// i.e. a token pass with per-image query rows u = g * w on the normalised tokens, an additive score bias w . pos_n (one
// small contraction), the mean row as one extra softmax entry (merged after the pass, as in the CaiT head), and the
// position-embedding part of the values as A . pos (another small contraction on the explicit attention weights).
// The passes run on ep_imgqf_kernel (ep_pool_imgq.hip): full-width per-image rows + score bias + explicit dS with every
// token read once for all heads; the generic token-pass kernels implement the same contract (one read per head) and are
// what the tests compare it with (ep_debug_force_generic_pool).
#include "ep_side.h"
#include "ep_headkernels.h"

namespace ep {

int token_image_stats(const void* x, int x_dtype, int64_t bstride, int B, int N, int D, float eps, int mode, float* stats, float* out,
                      hipStream_t st);   // ep_aim.hip

// xbar[b,:] = mean_n xhat[b,n,:],  xhat = (x - mean_n) rstd_n      (a thread owns 4 channels and walks the tokens)
template <bool BF16>
__global__ __launch_bounds__(256) void ep_xhat_mean_kernel(const void* __restrict__ x, int64_t bstride,
                                                         const int* __restrict__ index, const float* __restrict__ tokstat,
                                                         int N, int D, float* __restrict__ xbar) {
  const int b = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= D) return;
  const int64_t img = (int64_t)(index ? index[b] : b);
  const int64_t e0 = img * bstride + c;
  const float* ts = tokstat + img * N * 2;
  // four partial sums (n = w, w + 4, ...) combined as (p0 + p1) + (p2 + p3): the order of ep_tokimg_kernel (ep_aim.hip)
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  f4 s[4] = {z, z, z, z};
  int n0 = 0;
  for (; n0 + 8 <= N; n0 += 8) {                       // eight independent row loads in flight
    f4 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = load_tok4<BF16>(x, e0 + (int64_t)(n0 + t) * D);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float mean = ts[2 * (n0 + t)], rstd = ts[2 * (n0 + t) + 1];
      f4& a = s[t & 3];
      a.x = fmaf(__fsub_rn(v[t].x, mean), rstd, a.x); a.y = fmaf(__fsub_rn(v[t].y, mean), rstd, a.y);
      a.z = fmaf(__fsub_rn(v[t].z, mean), rstd, a.z); a.w = fmaf(__fsub_rn(v[t].w, mean), rstd, a.w);
    }
  }
  for (; n0 < N; n0 += 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (n0 + t < N) {
        const f4 v = load_tok4<BF16>(x, e0 + (int64_t)(n0 + t) * D);
        const float mean = ts[2 * (n0 + t)], rstd = ts[2 * (n0 + t) + 1];
        s[t].x = fmaf(__fsub_rn(v.x, mean), rstd, s[t].x); s[t].y = fmaf(__fsub_rn(v.y, mean), rstd, s[t].y);
        s[t].z = fmaf(__fsub_rn(v.z, mean), rstd, s[t].z); s[t].w = fmaf(__fsub_rn(v.w, mean), rstd, s[t].w);
      }
  }
  const float inv = 1.0f / (float)N;
  const f4 t = {__fadd_rn(__fadd_rn(s[0].x, s[1].x), __fadd_rn(s[2].x, s[3].x)), __fadd_rn(__fadd_rn(s[0].y, s[1].y), __fadd_rn(s[2].y, s[3].y)),
                __fadd_rn(__fadd_rn(s[0].z, s[1].z), __fadd_rn(s[2].z, s[3].z)), __fadd_rn(__fadd_rn(s[0].w, s[1].w), __fadd_rn(s[2].w, s[3].w))};
  *reinterpret_cast<f4*>(xbar + (int64_t)b * D + c) = f4{__fmul_rn(t.x, inv), __fmul_rn(t.y, inv), __fmul_rn(t.z, inv), __fmul_rn(t.w, inv)};
}

// t0 = g * xbar + b + pos_0
__global__ void ep_clip_t0_kernel(const float* __restrict__ xbar, const float* __restrict__ g, const float* __restrict__ beta,
                                  const float* __restrict__ pos0, int64_t total, int D, float* __restrict__ t0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  t0[i] = fmaf(g[d], xbar[i], beta[d] + pos0[d]);
}

// s0[r] = u[r,:] . xbar[b(r),:] + w[r,:] . pos_0        (r = b*H + h, one wave per row)
__global__ __launch_bounds__(256) void ep_clip_s0_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                       const float* __restrict__ xbar, const float* __restrict__ pos0, int rows,
                                                       int H, int D, float* __restrict__ s0) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63, b = r / H;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64)
    acc = fmaf(u[(int64_t)r * D + d], xbar[(int64_t)b * D + d], fmaf(w[(int64_t)r * D + d], pos0[d], acc));
  acc = wave_sum(acc);
  if (lane == 0) s0[r] = acc;
}

// merge the mean-row entry into every (image, head) softmax and write the explicit attention weights of the patch rows:
//   m' = max(m, s0) ; l' = l e^(m-m') + e^(s0-m') ; rho = l e^(m-m') / l' ; a0 = e^(s0-m') / l' ; A[n] = e^(S[n]-m') / l'
__global__ __launch_bounds__(256) void ep_clip_merge_kernel(const float* __restrict__ S, const float* __restrict__ ML,
                                                          const float* __restrict__ s0, int N, float* __restrict__ ML2,
                                                          float* __restrict__ mix, float* __restrict__ A) {
  const int r = blockIdx.x;
  const float m = ML[(int64_t)r * 4], l = ML[(int64_t)r * 4 + 1], s = s0[r];
  const float mn = fmaxf(m, s);
  const float et = l * expf(m - mn), ec = expf(s - mn);
  const float ln = et + ec, inv = 1.0f / ln;
  for (int n = threadIdx.x; n < N; n += 256) A[(int64_t)r * N + n] = expf(S[(int64_t)r * N + n] - mn) * inv;
  if (threadIdx.x == 0) {
    ML2[(int64_t)r * 4] = mn; ML2[(int64_t)r * 4 + 1] = ln;
    mix[r * 2] = et * inv; mix[r * 2 + 1] = ec * inv;
  }
}

// Ppr = rho Phat + a0 xbar ; vin = g * Ppr + (Apos + a0 pos_0) + b        (element-wise over (B*H, D))
__global__ void ep_clip_vin_kernel(const float* __restrict__ Ph, const float* __restrict__ xbar, const float* __restrict__ mix,
                                   const float* __restrict__ Apos, const float* __restrict__ g, const float* __restrict__ beta,
                                   const float* __restrict__ pos0, int64_t total, int D, int H, float* __restrict__ Ppr,
                                   float* __restrict__ vin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / D; const int d = (int)(i % D);
  const float rho = mix[r * 2], a0 = mix[r * 2 + 1];
  const float pp = fmaf(rho, Ph[i], a0 * xbar[(r / H) * D + d]);
  Ppr[i] = pp;
  vin[i] = fmaf(g[d], pp, fmaf(a0, pos0[d], Apos[i]) + beta[d]);
}

// column reductions over `rows` rows of width D (16 columns per workgroup; the rows split over gridDim.y chunks, fixed order):
//   o1[d] (+)= sum_r a[r,d] * (b ? b[brow(r),d] : 1) ;  o2[d] (+)= sum_r a[r,d] * wgt[r]   (o2 / wgt optional)
// brow(r) = r / bdiv (bdiv = H when b is a per-image table, 1 when it has one row per r).  With gridDim.y > 1 the chunk sums
// go to part[0 | 1][chunk][D] and clip_colred() finishes with ep_reduce_partials_kernel.
constexpr int CLIP_RS = 32;
__global__ __launch_bounds__(256) void ep_clip_colred_kernel(const float* __restrict__ a, const float* __restrict__ bm, int bdiv,
                                                           const float* __restrict__ wgt, int wstride, int rows, int D,
                                                           int acc1, float* __restrict__ o1, int acc2, float* __restrict__ o2,
                                                           float* __restrict__ part) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * CG + tx;
  const bool ok = c < D;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int r = r0 + ty; r < r1; r += RL) {
      const float v = a[(int64_t)r * D + c];
      s1 = bm ? fmaf(v, bm[(int64_t)(r / bdiv) * D + c], s1) : s1 + v;
      if (wgt) s2 = fmaf(v, wgt[(int64_t)r * wstride], s2);
    }
  s1 = colreduce(s1, sm, tx, ty);
  s2 = colreduce(s2, sm, tx, ty);
  if (ty == 0 && ok) {
    if (gridDim.y > 1) {
      part[(int64_t)blockIdx.y * D + c] = s1;
      part[((int64_t)gridDim.y + blockIdx.y) * D + c] = s2;
    } else {
      if (o1) o1[c] = acc1 ? o1[c] + s1 : s1;
      if (o2) o2[c] = acc2 ? o2[c] + s2 : s2;
    }
  }
}

static int clip_colred(const float* a, const float* bm, int bdiv, const float* wgt, int wstride, int rows, int D, int acc1, float* o1,
                       int acc2, float* o2, float* part, hipStream_t st) {
  int rs = rows / 128;
  rs = rs < 1 ? 1 : (rs > CLIP_RS ? CLIP_RS : rs);
  hipLaunchKernelGGL(ep_clip_colred_kernel, dim3((D + CG - 1) / CG, rs), dim3(256), 0, st, a, bm, bdiv, wgt, wstride, rows, D, acc1, o1,
                     acc2, o2, part);
  EP_LAUNCH_CHECK("ep_clip_colred_kernel");
  if (rs > 1) {
    if (o1) EP_TRY(reduce_partials(part, rs, D, 1.0f, acc1, o1, nullptr, st));
    if (o2) EP_TRY(reduce_partials(part + (size_t)rs * D, rs, D, 1.0f, acc2, o2, nullptr, st));
  }
  return 0;
}

// dPpr = dvin * g        (element-wise over (B*H, D))
__global__ void ep_clip_dppr_kernel(const float* __restrict__ dvin, const float* __restrict__ g, int64_t total, int D,
                                    float* __restrict__ dPpr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dPpr[i] = dvin[i] * g[i % D];
}

// per (image, head) -- one wave: the softmax correction of the (N + 1)-entry distribution and the mean-row entry's gradient
//   dA_0 = dPpr . xbar + dvin . pos_0 ;  delta' = dPpr . Ppr + sum_n A_n dAp_n + a0 (dvin . pos_0)   [dPpr . Ppr holds a0 dPpr . xbar]
//   dS_0 = a0 (dA_0 - delta') ;  ML2[r][2] = delta' ;  ds0[r] = dS_0
__global__ __launch_bounds__(256) void ep_clip_delta_kernel(const float* __restrict__ dPpr, const float* __restrict__ Ppr,
                                                          const float* __restrict__ dvin, const float* __restrict__ xbar,
                                                          const float* __restrict__ pos0, const float* __restrict__ A,
                                                          const float* __restrict__ dAp, const float* __restrict__ mix, int rows,
                                                          int H, int D, int N, float* __restrict__ ML2, float* __restrict__ ds0) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63, b = r / H;
  float pp = 0.f, px = 0.f, vp = 0.f, ad = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float g = dPpr[(int64_t)r * D + d];
    pp = fmaf(g, Ppr[(int64_t)r * D + d], pp);
    px = fmaf(g, xbar[(int64_t)b * D + d], px);
    vp = fmaf(dvin[(int64_t)r * D + d], pos0[d], vp);
  }
  for (int n = lane; n < N; n += 64) ad = fmaf(A[(int64_t)r * N + n], dAp[(int64_t)r * N + n], ad);
  pp = wave_sum(pp); px = wave_sum(px); vp = wave_sum(vp); ad = wave_sum(ad);
  if (lane == 0) {
    const float a0 = mix[r * 2 + 1];
    const float delta = pp + ad + a0 * vp;
    ML2[(int64_t)r * 4 + 2] = delta;
    ds0[r] = a0 * (px + vp - delta);
  }
}

// du = du0 + dS_0 xbar ;  dw = g * du + dwp + dS_0 pos_0       (element-wise over (B*H, D); du overwrites du0)
__global__ void ep_clip_dw_kernel(float* __restrict__ du, const float* __restrict__ dwp, const float* __restrict__ ds0,
                                  const float* __restrict__ xbar, const float* __restrict__ g, const float* __restrict__ pos0,
                                  int64_t total, int D, int H, float* __restrict__ dw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / D; const int d = (int)(i % D);
  const float s = ds0[r];
  const float v = fmaf(s, xbar[(r / H) * D + d], du[i]);
  du[i] = v;
  dw[i] = fmaf(g[d], v, fmaf(s, pos0[d], dwp[i]));
}

// ---------------------------------------------------------------------------------------------
constexpr int CLIP_NT = 9;    // pos_embed | qkv.weight qkv.bias | proj.weight proj.bias | norm.weight norm.bias | fc.weight fc.bias
struct ClipWs {
  float *tstat, *xbar, *t0, *q0, *w, *u, *sb, *s0, *P, *S, *ML, *ML2, *mix, *A, *Apos, *Ppr, *vin, *o;
  float *dO, *dvin, *dPpr, *dAp, *dS, *ds0, *du, *dwp, *dw, *dq0, *dt0, *cpart;
  float* skws; size_t skws_floats;                   // K-slice scratch of the two position-embedding gradients (gemm_split_k)
  void* pool_ws; size_t pool_ws_bytes;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static void clip_sizes(const ep_clip_dims& d, int64_t sizes[CLIP_NT]) {
  const int64_t D = d.D;
  const int64_t s[CLIP_NT] = {(int64_t)(d.N + 1) * D, 3 * D * D, 3 * D, D * D, D, D, D, (int64_t)d.C * D, d.C};
  for (int i = 0; i < CLIP_NT; ++i) sizes[i] = s[i];
}
static int64_t clip_offsets(const ep_clip_dims& d, int64_t offs[CLIP_NT]) {
  int64_t sizes[CLIP_NT];
  clip_sizes(d, sizes);
  int64_t off = 0;
  for (int i = 0; i < CLIP_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static ClipWs clip_carve(const ep_clip_dims& d, void* base, bool head) {
  ClipWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, H = d.H, N = d.N;
  w.tstat = take(B * N * 2); w.xbar = take(B * D); w.t0 = take(B * D); w.q0 = take(B * D);
  w.w = take(B * H * D); w.u = take(B * H * D); w.sb = take(B * H * N); w.s0 = take(B * H);
  w.P = take(B * H * D); w.S = take(B * H * N); w.ML = take(B * H * 4); w.ML2 = take(B * H * 4); w.mix = take(B * H * 2);
  w.A = take(B * H * N); w.Apos = take(B * H * D); w.Ppr = take(B * H * D); w.vin = take(B * H * D); w.o = take(B * D);
  w.dO = take(B * D); w.dvin = take(B * H * D); w.dPpr = take(B * H * D); w.dAp = take(B * H * N); w.dS = take(B * H * N);
  w.ds0 = take(B * H); w.du = take(B * H * D); w.dwp = take(B * H * D); w.dw = take(B * H * D); w.dq0 = take(B * D);
  w.dt0 = take(B * D); w.cpart = take((size_t)2 * CLIP_RS * D);
  w.skws_floats = (size_t)16 * N * D; w.skws = take(w.skws_floats);
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[CLIP_NT];
    w.opt_ws_bytes = optim_workspace_bytes(clip_offsets(d, offs), CLIP_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int clip_check(const ep_clip_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0, EP_E_ARG, "clip dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0, EP_E_SHAPE, "clip: D %% H == 0 and D/H, D multiples of 4");
  EP_REQUIRE(d.N % 4 == 0, EP_E_SHAPE, "clip: the token count N = %d (= feat_size^2) must be a multiple of 4", d.N);
  EP_REQUIRE(d.H <= 32, EP_E_UNSUPPORTED, "clip: more than 32 heads");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "clip head: C must be positive");
  return 0;
}

static int clip_params_ok(const ep_clip_params* p, const char* what) {
  EP_REQUIRE(p, EP_E_ARG, "%s: null parameter struct", what);
  const float* ts[] = {p->pos_embed, p->qkv_w, p->qkv_b, p->proj_w, p->proj_b, p->norm_w, p->norm_b};
  for (const float* t : ts) EP_REQUIRE(t && aligned16(t), EP_E_ALIGN, "%s: tensors must be non-null and 16-byte aligned", what);
  return 0;
}

static GemmParams kg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

int xhat_mean(const void* x, int x_dtype, int64_t bstride, const int32_t* index, const float* tokstat, int B, int N, int D,
              float* xbar, hipStream_t st) {
  const dim3 grid(B, (D / 4 + 255) / 256);
  if (x_dtype == EP_DTYPE_BF16)
    hipLaunchKernelGGL(ep_xhat_mean_kernel<true>, grid, dim3(256), 0, st, x, bstride, index, tokstat, N, D, xbar);
  else
    hipLaunchKernelGGL(ep_xhat_mean_kernel<false>, grid, dim3(256), 0, st, x, bstride, index, tokstat, N, D, xbar);
  EP_LAUNCH_CHECK("ep_xhat_mean_kernel");
  return 0;
}

static PoolParams clip_pool_params(const ep_clip_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                                   const float* tokstat, const ClipWs& w) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);
  p.cls = w.u; p.cls_bstride = (int64_t)d.H * d.D; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index; p.tokstat = tokstat;
  return p;
}

// rows of a cached per-image table: out[b, :] = table[index[b], :]
__global__ void ep_clip_gather_rows_kernel(const float* __restrict__ table, const int* __restrict__ index, int64_t total, int rowlen,
                                           float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = table[(int64_t)(index ? index[i / rowlen] : (int)(i / rowlen)) * rowlen + (i % rowlen)];
}

static int clip_forward_core(const ep_clip_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const float* tokstat, const ep_clip_params& pr, const ClipWs& w, float* y, hipStream_t st,
                             const float* xbar_table = nullptr) {
  const int D = d.D, dh = D / d.H, B = d.B, H = d.H, N = d.N, BH = B * H;
  const float scale = (float)pow((double)dh, -0.5);                        // attention_pool2d.py:124
  const float* pos0 = pr.pos_embed; const float* posN = pr.pos_embed + D;  // row 0: the mean row; rows 1..N: the patch rows
  const float* Wq = pr.qkv_w; const float* Wk = pr.qkv_w + (int64_t)D * D; const float* Wv = pr.qkv_w + 2 * (int64_t)D * D;
  bool have_xbar = false;
  if (!tokstat) {
    EP_REQUIRE(!index, EP_E_ARG, "clip: an indexed token store needs precomputed token statistics");
    // statistics and the mean normalised row in ONE read of the batch (same bits as the two separate kernels)
    have_xbar = (D + 255) / 256 <= 5;
    if (have_xbar) EP_TRY(token_image_stats(x, x_dtype, bstride, B, N, D, d.ln_eps, 1, w.tstat, w.xbar, st));
    else EP_TRY(token_stats(x, x_dtype == EP_DTYPE_BF16, bstride, B, N, D, d.ln_eps, w.tstat, st));
    tokstat = w.tstat;
  }
  if (!have_xbar && xbar_table) {                    // the store's cached table (ABI v23): the batch's rows, no token read
    const int64_t nbd = (int64_t)B * D;
    hipLaunchKernelGGL(ep_clip_gather_rows_kernel, dim3((unsigned)((nbd + 255) / 256)), dim3(256), 0, st, xbar_table, index, nbd, D, w.xbar);
    EP_LAUNCH_CHECK("ep_clip_gather_rows_kernel");
    have_xbar = true;
  }
  if (!have_xbar) EP_TRY(xhat_mean(x, x_dtype, bstride, index, tokstat, B, N, D, w.xbar, st));
  const int64_t nd = (int64_t)B * D, nhd = (int64_t)BH * D;
  hipLaunchKernelGGL(ep_clip_t0_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, w.xbar, pr.norm_w, pr.norm_b, pos0, nd,
                     D, w.t0);
  EP_LAUNCH_CHECK("ep_clip_t0_kernel");
  { GemmParams g = kg(w.t0, D, Wq, D, w.q0, D, B, D, D); g.bias = pr.qkv_b; EP_TRY(gemm(true, true, g, 1, st)); }     // q0 = t0 Wq^T + bq
  {
    GemmParams g = kg(w.q0, D, Wk, D, w.w, (int64_t)H * D, B, D, dh);                                               // w_h = scale q0_h Wk_h
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D; g.alpha = scale;
    EP_TRY(gemm(true, false, g, H, st));
  }
  hipLaunchKernelGGL(ep_rowscale_kernel, dim3((unsigned)((nhd + 255) / 256)), dim3(256), 0, st, w.w, pr.norm_w, BH, D, w.u);
  EP_TRY(gemm(true, true, kg(w.w, D, posN, D, w.sb, N, BH, N, D), 1, st));                                          // sb = w pos^T
  hipLaunchKernelGGL(ep_clip_s0_kernel, dim3((BH + 3) / 4), dim3(256), 0, st, w.u, w.w, w.xbar, pos0, BH, H, D, w.s0);
  EP_LAUNCH_CHECK("ep_clip query kernels");
  PoolParams p = clip_pool_params(d, x, x_dtype, bstride, index, tokstat, w);
  p.sbias = w.sb;
  EP_TRY(pool_forward(p, st));
  hipLaunchKernelGGL(ep_clip_merge_kernel, dim3(BH), dim3(256), 0, st, w.S, w.ML, w.s0, N, w.ML2, w.mix, w.A);
  EP_LAUNCH_CHECK("ep_clip_merge_kernel");
  { GemmParams g = kg(w.A, N, posN, D, w.Apos, D, BH, D, N); g.extB = D; EP_TRY(gemm(true, false, g, 1, st)); }       // Apos = A pos[1:]
  hipLaunchKernelGGL(ep_clip_vin_kernel, dim3((unsigned)((nhd + 255) / 256)), dim3(256), 0, st, w.P, w.xbar, w.mix, w.Apos,
                     pr.norm_w, pr.norm_b, pos0, nhd, D, H, w.Ppr, w.vin);
  EP_LAUNCH_CHECK("ep_clip_vin_kernel");
  {
    GemmParams g = kg(w.vin, (int64_t)H * D, Wv, D, w.o, D, B, dh, D);                                               // o_h = vin_h Wv_h^T + bv_h
    g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh; g.bias = pr.qkv_b + 2 * D; g.sBiasz = dh;
    EP_TRY(gemm(true, true, g, H, st));
  }
  GemmParams g = kg(w.o, D, pr.proj_w, D, y, D, B, D, D); g.bias = pr.proj_b;
  return gemm(true, true, g, 1, st);
}

static int clip_backward_core(const ep_clip_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                              const float* tokstat, const ep_clip_params& pr, const float* dy, const ep_clip_params& gr, int acc,
                              const ClipWs& w, hipStream_t st, AuxSide* axp = nullptr) {
  const int D = d.D, dh = D / d.H, B = d.B, H = d.H, N = d.N, BH = B * H;
  const float scale = (float)pow((double)dh, -0.5);
  const float* pos0 = pr.pos_embed; const float* posN = pr.pos_embed + D;
  const float* Wq = pr.qkv_w; const float* Wk = pr.qkv_w + (int64_t)D * D; const float* Wv = pr.qkv_w + 2 * (int64_t)D * D;
  float* dWq = gr.qkv_w; float* dWk = gr.qkv_w + (int64_t)D * D; float* dWv = gr.qkv_w + 2 * (int64_t)D * D;
  float* dpos0 = gr.pos_embed; float* dposN = gr.pos_embed + D;
  if (!tokstat) tokstat = w.tstat;
  const int64_t nhd = (int64_t)BH * D;
  const unsigned eh = (unsigned)((nhd + 255) / 256), cgrid = (D + CG - 1) / CG;
  // The six parameter-gradient contractions feed nothing before the optimizer: in the fused step they go to the aux stream, each
  // as soon as its operands exist (AuxSide, ep_internal.h); the caller joins.  Stand-alone: inline, in program order.
  AuxSide inline_ax;
  if (!axp) { EP_TRY(aux_side_begin(inline_ax, st, nullptr)); axp = &inline_ax; }
  AuxSide& ax = *axp;
  // y = o Wp^T + bp
  EP_TRY(colsum(dy, B, D, D, acc, gr.proj_b, st));
  { GemmParams g = kg(dy, D, w.o, D, gr.proj_w, D, D, D, B); g.accumulate = acc; EP_TRY(aux_side_gemm(ax, g, 1)); }
  EP_TRY(gemm(true, false, kg(dy, D, pr.proj_w, D, w.dO, D, B, D, D), 1, st));                                        // dO = dy Wp
  // o_h = vin_h Wv_h^T + bv_h
  EP_TRY(colsum(w.dO, B, D, D, acc, gr.qkv_b + 2 * D, st));
  {
    GemmParams g = kg(w.dO, D, w.vin, (int64_t)H * D, dWv, D, dh, D, B);                                              // dWv_h = dO_h^T vin_h
    g.sAz = dh; g.extA = dh; g.sBz = D; g.extB = D; g.sCz = (int64_t)dh * D; g.accumulate = acc;
    EP_TRY(aux_side_gemm(ax, g, H));
  }
  {
    GemmParams g = kg(w.dO, D, Wv, D, w.dvin, (int64_t)H * D, B, D, dh);                                              // dvin_h = dO_h Wv_h
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D;
    EP_TRY(gemm(true, false, g, H, st));
  }
  // vin = g * Ppr + Apos + a0 pos_0 + b :  d g = sum dvin Ppr ; d b = sum dvin ; d pos_0 = sum a0 dvin ; dPpr = dvin g
  EP_TRY(clip_colred(w.dvin, w.Ppr, 1, (const float*)nullptr, 0, BH, D, acc,
                     gr.norm_w, 0, (float*)nullptr, w.cpart, st));
  EP_TRY(clip_colred(w.dvin, (const float*)nullptr, 1, w.mix + 1, 2, BH, D, acc,
                     gr.norm_b, acc, dpos0, w.cpart, st));
  hipLaunchKernelGGL(ep_clip_dppr_kernel, dim3(eh), dim3(256), 0, st, w.dvin, pr.norm_w, nhd, D, w.dPpr);
  EP_LAUNCH_CHECK("ep_clip value backward kernels");
  // Apos = A pos[1:] :  d pos[1:] = A^T dvin ; dAp = dvin pos[1:]^T
  // (d pos[1:]: N x D outputs -- 48 tiles -- summed over all B H rows: K slices, or one long latency chain on a fifth of the chip.
  // Round 6: the two of them took 153 + 128 us on the side queue and the optimizer waited 140 us for it, rocprofv3.)
  { GemmParams g = kg(w.A, N, w.dvin, D, dposN, D, N, D, BH); g.accumulate = acc; g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(aux_side_gemm(ax, g, 1)); }
  EP_TRY(gemm(true, true, kg(w.dvin, D, posN, D, w.dAp, N, BH, N, D), 1, st));
  hipLaunchKernelGGL(ep_clip_delta_kernel, dim3((BH + 3) / 4), dim3(256), 0, st, w.dPpr, w.Ppr, w.dvin, w.xbar, pos0, w.A, w.dAp,
                     w.mix, BH, H, D, N, w.ML2, w.ds0);
  EP_LAUNCH_CHECK("ep_clip_delta_kernel");
  // second token pass: dS (explicit) and the per-image query gradients of the patch rows
  PoolParams p = clip_pool_params(d, x, x_dtype, bstride, index, tokstat, w);
  p.ML = w.ML2; p.dP = w.dPpr; p.dabias = w.dAp; p.dSout = w.dS;
  EP_TRY(pool_backward_per_image(p, w.du, st));
  // dw = g * (du0 + dS_0 xbar) + dS pos[1:] + dS_0 pos_0 ;  d g += sum du w ;  d pos[1:] += dS^T w ; d pos_0 += sum dS_0 w
  { GemmParams g = kg(w.dS, N, posN, D, w.dwp, D, BH, D, N); g.extB = D; EP_TRY(gemm(true, false, g, 1, st)); }
  hipLaunchKernelGGL(ep_clip_dw_kernel, dim3(eh), dim3(256), 0, st, w.du, w.dwp, w.ds0, w.xbar, pr.norm_w, pos0, nhd, D, H, w.dw);
  EP_TRY(clip_colred(w.du, w.w, 1, (const float*)nullptr, 0, BH, D, 1,
                     gr.norm_w, 0, (float*)nullptr, w.cpart, st));
  EP_TRY(clip_colred(w.w, (const float*)nullptr, 1, w.ds0, 1, BH, D, 0,
                     (float*)nullptr, 1, dpos0, w.cpart, st));
  EP_LAUNCH_CHECK("ep_clip key backward kernels");
  { GemmParams g = kg(w.dS, N, w.w, D, dposN, D, N, D, BH); g.accumulate = 1; g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(aux_side_gemm(ax, g, 1)); }   // (behind the first dposN term: same stream, so the scratch is free again)
  // w_h = scale q0_h Wk_h :  dq0_h = scale dw_h Wk_h^T ; dWk_h = scale q0_h^T dw_h ; d bk = 0
  {
    GemmParams g = kg(w.dw, (int64_t)H * D, Wk, D, w.dq0, D, B, dh, D);
    g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh; g.alpha = scale;
    EP_TRY(gemm(true, true, g, H, st));
  }
  {
    GemmParams g = kg(w.q0, D, w.dw, (int64_t)H * D, dWk, D, dh, D, B);
    g.sAz = dh; g.extA = dh; g.sBz = D; g.extB = D; g.sCz = (int64_t)dh * D; g.alpha = scale; g.accumulate = acc;
    EP_TRY(aux_side_gemm(ax, g, H));
  }
  if (!acc) EP_HIP(hipMemsetAsync(gr.qkv_b + D, 0, (size_t)D * sizeof(float), st));
  // q0 = t0 Wq^T + bq ;  t0 = g * xbar + b + pos_0
  EP_TRY(colsum(w.dq0, B, D, D, acc, gr.qkv_b, st));
  { GemmParams g = kg(w.dq0, D, w.t0, D, dWq, D, D, D, B); g.accumulate = acc; EP_TRY(aux_side_gemm(ax, g, 1)); }
  EP_TRY(gemm(true, false, kg(w.dq0, D, Wq, D, w.dt0, D, B, D, D), 1, st));                                          // dt0 = dq0 Wq
  EP_TRY(clip_colred(w.dt0, w.xbar, 1, (const float*)nullptr, 0, B, D, 1,
                     gr.norm_w, 0, (float*)nullptr, w.cpart, st));
  EP_TRY(clip_colred(w.dt0, (const float*)nullptr, 1, (const float*)nullptr, 0,
                     B, D, 1, gr.norm_b, 0, (float*)nullptr, w.cpart, st));
  EP_TRY(colsum(w.dt0, B, D, D, 1, dpos0, st));
  EP_LAUNCH_CHECK("ep_clip query backward kernels");
  return 0;
}

static ep_clip_params clip_views(float* base, const int64_t o[CLIP_NT]) {
  ep_clip_params p;
  p.pos_embed = base + o[0]; p.qkv_w = base + o[1]; p.qkv_b = base + o[2]; p.proj_w = base + o[3]; p.proj_b = base + o[4];
  p.norm_w = base + o[5]; p.norm_b = base + o[6];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

int ep_token_xhat_mean(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, const float* token_stats_,
                       int B, int N, int D, float* xbar, ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, 1));
  EP_REQUIRE(token_stats_ && xbar && aligned16(xbar), EP_E_ARG, "ep_token_xhat_mean: null / unaligned pointer");
  return xhat_mean(x, x_dtype, x_bstride, image_index, token_stats_, B, N, D, xbar, (hipStream_t)stream);
}

size_t ep_clip_pool_workspace_bytes(const ep_clip_dims* dims) {
  if (!dims || clip_check(*dims, false) != 0) return 0;
  return clip_carve(*dims, nullptr, false).total;
}

int ep_clip_pool_forward(const ep_clip_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                         const float* token_stats_, const ep_clip_params* params, float* y, void* ws, size_t ws_bytes,
                         ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_clip_pool_forward: null pointer");
  EP_TRY(clip_check(*dims, false));
  EP_TRY(clip_params_ok(params, "ep_clip_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_clip_pool_forward: y / ws must be 16-byte aligned");
  const ClipWs w = clip_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_clip_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return clip_forward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, *params, w, y, (hipStream_t)stream);
}

int ep_clip_pool_backward(const ep_clip_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                          const float* token_stats_, const ep_clip_params* params, const float* dy,
                          const ep_clip_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dy && ws, EP_E_ARG, "ep_clip_pool_backward: null pointer");
  EP_TRY(clip_check(*dims, false));
  EP_TRY(clip_params_ok(params, "ep_clip_pool_backward(params)"));
  EP_TRY(clip_params_ok(grads, "ep_clip_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dy), EP_E_ALIGN, "ep_clip_pool_backward: dy / ws must be 16-byte aligned");
  const ClipWs w = clip_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_clip_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return clip_backward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, *params, dy, *grads, accumulate, w,
                            (hipStream_t)stream);
}

/* attention of the mean-row query over the patch rows, (B, H, N), of the last forward on this workspace
 * (attention_pool2d.py:167 return_attn: attn[:, :, 0, 1:]) */
int ep_clip_attention(const ep_clip_dims* dims, const void* ws, float* A, ep_stream_t stream) {
  EP_REQUIRE(dims && ws && A, EP_E_ARG, "ep_clip_attention: null pointer");
  EP_TRY(clip_check(*dims, false));
  const ClipWs w = clip_carve(*dims, const_cast<void*>(ws), false);
  EP_HIP(hipMemcpyAsync(A, w.A, (size_t)dims->B * dims->H * dims->N * sizeof(float), hipMemcpyDeviceToDevice,
                        (hipStream_t)stream));
  return 0;
}

int64_t ep_clip_head_param_offsets(const ep_clip_dims* dims, int64_t offsets[9]) { return clip_offsets(*dims, offsets); }

size_t ep_clip_head_workspace_bytes(const ep_clip_dims* dims) {
  if (!dims || clip_check(*dims, true) != 0) return 0;
  return clip_carve(*dims, nullptr, true).total;
}

int ep_clip_head_train_step(const ep_clip_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_clip_head_train_step: null pointer");
  const ep_clip_dims& d = s->dims;
  EP_TRY(clip_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const ClipWs w = clip_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_clip_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CLIP_NT];
  const int64_t total = clip_offsets(d, offs);
  const ep_clip_params pr = clip_views(s->params, offs), gr = clip_views(s->grads, offs);
  float* Wc = s->params + offs[7]; float* bc = s->params + offs[8];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    EP_REQUIRE(!s->xhat_mean || s->token_stats, EP_E_ARG, "clip: the xhat_mean table goes with the token_stats table it was made from");
    EP_TRY(clip_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, pr, w, w.y, st, s->xhat_mean));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    AuxSide ax;
    EP_TRY(aux_side_begin(ax, st, (hipStream_t)s->aux_stream));
    EP_TRY(classifier_backward(ax, w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, s->grads + offs[7], s->grads + offs[8],
                               s->accumulate));     // (the weight gradient: beside dz and the BatchNorm backward)
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    EP_TRY(clip_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, pr, w.dy, gr, s->accumulate, w,
                              st, &ax));
    EP_TRY(aux_side_join(ax));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    int64_t sizes[CLIP_NT];
    clip_sizes(d, sizes);
    const int trust[CLIP_NT] = {1, 1, 0, 1, 0, 0, 0, 1, 0};      // util/lars.py:22: ndim > 1 (pos_embed is (N + 1, D))
    ep_segment segs[CLIP_NT];
    for (int i = 0; i < CLIP_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? CLIP_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_clip_head_eval_forward(const ep_clip_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                              const float* token_stats_, const float* params, const float* running_mean,
                              const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                              ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_clip_head_eval_forward: null pointer");
  const ep_clip_dims& d = *dims;
  EP_TRY(clip_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const ClipWs w = clip_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_clip_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CLIP_NT];
  clip_offsets(d, offs);
  const ep_clip_params pr = clip_views(const_cast<float*>(params), offs);
  EP_TRY(clip_forward_core(d, x, x_dtype, x_bstride, image_index, token_stats_, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[7], params + offs[8], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
