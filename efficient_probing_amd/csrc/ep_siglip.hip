// SigLIP attention-pool head (reference poolings/clip/attention_pool.py:13-140 AttentionPoolLatent with the
// registry's arguments probe_heads.py:72: 8 heads, one latent query, qkv bias, mlp_ratio 4, no norms, pool
// 'token') on the EP streaming kernels.
//
// One learned latent query, projected by `q`; keys / values are per-head slices of kv(x).  Per head h:
//     score[b,h,n] = scale q_h . (Wk_h x[b,n] + bk_h) = u_h . x[b,n] + const        u_h = scale Wk_h^T q_h
//     o[b,h]       = sum_n A[b,h,n] (Wv_h x[b,n] + bv_h) = P[b,h] Wv_h^T + bv_h      (sum_n A = 1)
// (the key bias shifts every score of a head equally and cancels in the softmax), so the token-dependent part is
// the EP pooling pass with H derived query rows u and the EP per-query value projection with Wv = kv.weight[D:],
// followed per image by proj and the residual MLP:  z1 = o Wp^T + bp;  out = z1 + W2 gelu(W1 z1 + b1) + b2.
// The backward needs du only; the chain from du to kv.weight[:D], q.weight / bias and the latent is batch
// independent.  d kv.bias[:D] is exactly zero.
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_lnaffine.h"
#include "ep_headkernels.h"

namespace ep {

// ---------------------------------------------------------------------------------------------
struct SigWs {
  float *P, *S, *ML, *ya, *z1, *pre, *h1, *dh1, *dz1, *dya, *dP, *q, *u, *du, *dq;
  float* skws; size_t skws_floats;                   // K-slice scratch of the two long-K MLP contractions (ep_gemm.hip: gemm_split_k)
  void* pool_ws; size_t pool_ws_bytes;
  size_t pool_total;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

constexpr int SIG_NT = 13;     // latent q.w q.b kv.w kv.b proj.w proj.b fc1.w fc1.b fc2.w fc2.b | fc.weight fc.bias

static int64_t sig_offsets(const ep_siglip_dims& d, int64_t offs[SIG_NT]) {
  const int64_t D = d.D, Hd = d.hidden;
  const int64_t sizes[SIG_NT] = {D, D * D, D, 2 * D * D, 2 * D, D * D, D, Hd * D, Hd, D * Hd, D, (int64_t)d.C * D, d.C};
  int64_t off = 0;
  for (int i = 0; i < SIG_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static SigWs sig_carve(const ep_siglip_dims& d, void* base, bool head) {
  SigWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, Hd = d.hidden;
  w.P = take(B * d.H * D); w.S = take(B * d.H * d.N); w.ML = take(B * d.H * 4);
  w.ya = take(B * D); w.z1 = take(B * D); w.pre = take(B * Hd); w.h1 = take(B * Hd); w.dh1 = take(B * Hd);
  w.dz1 = take(B * D); w.dya = take(B * D); w.dP = take(B * d.H * D);
  w.q = take(D); w.u = take((size_t)d.H * D); w.du = take((size_t)d.H * D); w.dq = take(D);
  w.skws_floats = 4 * B * D; w.skws = take(w.skws_floats);
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.H);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  w.pool_total = off;
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[SIG_NT];
    w.opt_ws_bytes = optim_workspace_bytes(sig_offsets(d, offs), SIG_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int sig_check(const ep_siglip_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0 && d.hidden > 0, EP_E_ARG, "siglip dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0 && d.hidden % 4 == 0, EP_E_SHAPE,
             "siglip: D %% H == 0 and D/H, D, hidden multiples of 4 (D=%d H=%d hidden=%d)", d.D, d.H, d.hidden);
  EP_REQUIRE(d.H <= 32, EP_E_UNSUPPORTED, "siglip: heads = %d > 32", d.H);
  EP_REQUIRE((size_t)(2 * d.D + 256) * 4 <= 60000, EP_E_UNSUPPORTED, "siglip: D too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "siglip head: C must be positive");
  return 0;
}

static int sig_params_ok(const ep_siglip_params* p, const char* what) {
  EP_REQUIRE(p && p->latent && p->q_w && p->q_b && p->kv_w && p->kv_b && p->proj_w && p->proj_b && p->fc1_w && p->fc1_b &&
             p->fc2_w && p->fc2_b, EP_E_ARG, "%s: null tensor", what);
  EP_REQUIRE(aligned16(p->latent) && aligned16(p->q_w) && aligned16(p->q_b) && aligned16(p->kv_w) && aligned16(p->kv_b) &&
             aligned16(p->proj_w) && aligned16(p->proj_b) && aligned16(p->fc1_w) && aligned16(p->fc1_b) && aligned16(p->fc2_w) &&
             aligned16(p->fc2_b), EP_E_ALIGN, "%s: tensors must be 16-byte aligned", what);
  return 0;
}

static PoolParams sig_pool_params(const ep_siglip_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                                  const SigWs& w) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);      // the scale lives in u
  p.cls = w.u; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index;
  return p;
}

static GemmParams mkg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

static int sig_forward_core(const ep_siglip_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                            const ep_siglip_params& pr, const SigWs& w, float* out, hipStream_t st) {
  const int D = d.D, dh = D / d.H, Hd = d.hidden;
  const float scale = (float)pow((double)dh, -0.5);                        // attention_pool.py:40
  hipLaunchKernelGGL(ep_siglip_q_kernel, dim3((D + 3) / 4), dim3(256), 0, st, pr.latent, pr.q_w, pr.q_b, D, w.q);
  EP_TRY(siglip_u(w.q, pr.kv_w, D, d.H, dh, scale, w.u, st));
  EP_LAUNCH_CHECK("ep_siglip_q/u kernels");
  EP_TRY(pool_forward(sig_pool_params(d, x, x_dtype, bstride, index, w), st));
  {
    GemmParams g = mkg(w.P, (int64_t)d.H * D, pr.kv_w + (int64_t)D * D, D, w.ya, D, d.B, dh, D);   // o = P Wv_h^T + bv_h
    g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh; g.bias = pr.kv_b + D; g.sBiasz = dh;
    EP_TRY(gemm(true, true, g, d.H, st));
  }
  { GemmParams g = mkg(w.ya, D, pr.proj_w, D, w.z1, D, d.B, D, D); g.bias = pr.proj_b; EP_TRY(gemm(true, true, g, 1, st)); }
  { GemmParams g = mkg(w.z1, D, pr.fc1_w, D, w.pre, Hd, d.B, Hd, D); g.bias = pr.fc1_b; EP_TRY(gemm(true, true, g, 1, st)); }
  const int64_t n4 = (int64_t)d.B * Hd / 4;
  hipLaunchKernelGGL(ep_gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.h1);
  EP_LAUNCH_CHECK("ep_gelu_kernel");
  EP_HIP(hipMemcpyAsync(out, w.z1, (size_t)d.B * D * sizeof(float), hipMemcpyDeviceToDevice, st));      // residual
  { GemmParams g = mkg(w.h1, Hd, pr.fc2_w, Hd, out, D, d.B, D, Hd); g.bias = pr.fc2_b; g.accumulate = 1;
    g.skws = w.skws; g.skws_floats = w.skws_floats; EP_TRY(gemm(true, true, g, 1, st)); }
  return 0;
}

static int sig_backward_core(const ep_siglip_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const ep_siglip_params& pr, const float* dout, const ep_siglip_params& gr, int acc,
                             const SigWs& w, SideTasks sd, hipStream_t st, hipStream_t aux) {
  const int D = d.D, dh = D / d.H, Hd = d.hidden, B = d.B;
  const float scale = (float)pow((double)dh, -0.5);
  const int64_t n4 = (int64_t)B * Hd / 4;
  // Weight and bias gradients: nothing consumes them before the optimizer.  They ride in the second token pass as side
  // workgroups where its kernel takes them; otherwise (or with EP_SIDE_AUX=1) on the aux stream, each contraction as early as
  // its operands exist (AuxSide, ep_internal.h).
  PoolParams p = sig_pool_params(d, x, x_dtype, bstride, index, w);
  p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
  static int aux_env = -1;
  if (aux_env < 0) { const char* e = getenv("EP_SIDE_AUX"); aux_env = e ? atoi(e) : 0; }
  const bool in_pass = pool_backward_takes_side(p) && !(aux_env && aux && aux != st);
  AuxSide ax;
  EP_TRY(aux_side_begin(ax, st, aux));
  auto fork = [&]() -> int { return in_pass ? 0 : aux_side_fork(ax, sd); };
  GemmParams gW2 = mkg(dout, D, w.h1, Hd, gr.fc2_w, Hd, D, Hd, B); gW2.accumulate = acc; gW2.side = 1;       // dW2 = dout^T h1
  GemmParams gW1 = mkg(w.dh1, Hd, w.z1, D, gr.fc1_w, D, Hd, D, B); gW1.accumulate = acc; gW1.side = 1;       // dW1 = dpre^T z1
  GemmParams gWp = mkg(w.dz1, D, w.ya, D, gr.proj_w, D, D, D, B); gWp.accumulate = acc; gWp.side = 1;        // dWp = dz1^T o
  GemmParams gWv = mkg(w.dya, D, w.P, (int64_t)d.H * D, gr.kv_w + (int64_t)D * D, D, dh, D, B);              // dWv_h = dya_h^T P_h
  gWv.sAz = dh; gWv.extA = dh; gWv.sBz = D; gWv.extB = D; gWv.sCz = (int64_t)dh * D; gWv.accumulate = acc; gWv.side = 1;
  EP_REQUIRE(gemm_side_ok(gW2, false, false) && gemm_side_ok(gW1, false, false) && gemm_side_ok(gWp, false, false) &&
             gemm_side_ok(gWv, false, false), EP_E_ALIGN, "siglip: unaligned gradient contraction");
  side_add_gemm(sd, gW2, 1);
  EP_TRY(fork());
  // MLP: out = z1 + fc2(gelu(fc1(z1)))
  EP_TRY(gemm(true, false, mkg(dout, D, pr.fc2_w, Hd, w.dh1, Hd, B, Hd, D), 1, st));               // dh1 = dout W2
  hipLaunchKernelGGL(ep_gelu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.dh1);   // -> dpre
  EP_LAUNCH_CHECK("ep_gelu_bwd_kernel");
  side_add_gemm(sd, gW1, 1);
  EP_TRY(fork());
  EP_HIP(hipMemcpyAsync(w.dz1, dout, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, st));
  { GemmParams g = mkg(w.dh1, Hd, pr.fc1_w, D, w.dz1, D, B, D, Hd); g.accumulate = 1; g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(true, false, g, 1, st)); }   // dz1 = dout + dpre W1
  side_add_gemm(sd, gWp, 1);
  EP_TRY(fork());
  EP_TRY(gemm(true, false, mkg(w.dz1, D, pr.proj_w, D, w.dya, D, B, D, D), 1, st));                 // dya = dz1 Wp
  side_add_gemm(sd, gWv, d.H);
  EP_TRY(fork());
  // bias gradients (column sums over the batch)
  if (!side_add_colsum(sd, dout, B, D, D, acc, gr.fc2_b)) EP_TRY(colsum(dout, B, D, D, acc, gr.fc2_b, st));
  if (!side_add_colsum(sd, w.dh1, B, Hd, Hd, acc, gr.fc1_b)) EP_TRY(colsum(w.dh1, B, Hd, Hd, acc, gr.fc1_b, st));
  if (!side_add_colsum(sd, w.dz1, B, D, D, acc, gr.proj_b)) EP_TRY(colsum(w.dz1, B, D, D, acc, gr.proj_b, st));
  if (!side_add_colsum(sd, w.dya, B, D, D, acc, gr.kv_b + D)) EP_TRY(colsum(w.dya, B, D, D, acc, gr.kv_b + D, st));
  if (!in_pass) EP_TRY(aux_side_rest(ax, sd));
  EP_TRY(delta_rows(w.dya, w.ya, B * d.H, dh, w.ML, st, pr.kv_b + D, d.H));                          // dP . P (bias taken out)
  {
    GemmParams g = mkg(w.dya, D, pr.kv_w + (int64_t)D * D, D, w.dP, (int64_t)d.H * D, B, D, dh);     // dP[b,h] = dya[b,h] Wv_h
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D;
    EP_TRY(gemm(true, false, g, d.H, st));
  }
  if (in_pass) {
    EP_TRY(pool_backward(p, w.du, 0, st, &sd));
  } else {
    EP_TRY(aux_side_before_pass(ax, sd));
    EP_TRY(pool_backward(p, w.du, 0, st));
    EP_TRY(aux_side_join(ax));
  }
  // query chain
  hipLaunchKernelGGL(ep_siglip_dq_kernel, dim3((D + 3) / 4), dim3(256), 0, st, w.du, pr.kv_w, D, dh, scale, acc, w.dq, gr.q_b);
  EP_TRY(siglip_qgrad(w.q, w.dq, w.du,
                     pr.latent, pr.q_w, D, dh, scale, acc, gr.kv_w, gr.q_w, gr.latent, gr.kv_b, st));
  EP_LAUNCH_CHECK("ep_siglip backward kernels");
  return 0;
}

static ep_siglip_params sig_views(float* base, const int64_t o[SIG_NT]) {
  ep_siglip_params p;
  p.latent = base + o[0]; p.q_w = base + o[1]; p.q_b = base + o[2]; p.kv_w = base + o[3]; p.kv_b = base + o[4];
  p.proj_w = base + o[5]; p.proj_b = base + o[6]; p.fc1_w = base + o[7]; p.fc1_b = base + o[8]; p.fc2_w = base + o[9];
  p.fc2_b = base + o[10];
  return p;
}


// =============================================================================================
// V-JEPA attentive pooler (reference poolings/jepa/attentive_pooler.py:21-104 with CrossAttentionBlock / CrossAttention /
// MLP of poolings/jepa/modules.py:13-183; registry entry probe_heads.py:81: AttentivePooler(embed_dim=dim,
// num_heads=args.num_heads), one query token, depth 1, complete block):
//     y  = xattn(q0, LN1(x));  q1 = q0 + y;  out = q1 + mlp(LN2(q1))
// = the SigLIP head above with (a) the keys / values taken from LayerNorm-ed tokens (LayerNorm-of-tokens mode of the
// passes, the affine part folded into the query rows and the value projection as in the CAE head), (b) the residual with
// the query token and (c) a LayerNorm in front of the MLP.
// =============================================================================================
constexpr int JEPA_NT = 17;   // query | n1.w n1.b | q.w q.b | kv.w kv.b | proj.w proj.b | n2.w n2.b | fc1.w fc1.b | fc2.w fc2.b | fc.w fc.b
struct JepaWs {
  float *P, *S, *ML, *tstat, *ya, *q1, *qstat, *h2, *pre, *h1, *dh1, *dh2, *dq1, *dya, *dP, *q, *u, *wq, *dw, *du, *dq, *Wvs, *bo,
      *dWvs, *dbo, *bq0;
  float* skws; size_t skws_floats;                   // (as SigWs)
  void* pool_ws; size_t pool_ws_bytes;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static int64_t jepa_offsets(const ep_jepa_dims& d, int64_t offs[JEPA_NT]) {
  const int64_t D = d.D, Hd = d.hidden;
  const int64_t sizes[JEPA_NT] = {D, D, D, D * D, D, 2 * D * D, 2 * D, D * D, D, D, D, Hd * D, Hd, D * Hd, D, (int64_t)d.C * D, d.C};
  int64_t off = 0;
  for (int i = 0; i < JEPA_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static JepaWs jepa_carve(const ep_jepa_dims& d, void* base, bool head) {
  JepaWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, Hd = d.hidden;
  w.P = take(B * d.H * D); w.S = take(B * d.H * d.N); w.ML = take(B * d.H * 4); w.tstat = take(B * d.N * 2);
  w.ya = take(B * D); w.q1 = take(B * D); w.qstat = take(B * 2); w.h2 = take(B * D); w.pre = take(B * Hd); w.h1 = take(B * Hd);
  w.dh1 = take(B * Hd); w.dh2 = take(B * D); w.dq1 = take(B * D); w.dya = take(B * D); w.dP = take(B * d.H * D);
  w.q = take(D); w.u = take((size_t)d.H * D); w.wq = take((size_t)d.H * D); w.dw = take((size_t)d.H * D);
  w.du = take((size_t)d.H * D); w.dq = take(D); w.Wvs = take(D * D); w.bo = take(D); w.dWvs = take(D * D); w.dbo = take(D);
  w.bq0 = take(D);
  w.skws_floats = 4 * B * D; w.skws = take(w.skws_floats);
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.H);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[JEPA_NT];
    w.opt_ws_bytes = optim_workspace_bytes(jepa_offsets(d, offs), JEPA_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int jepa_check(const ep_jepa_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0 && d.hidden > 0, EP_E_ARG, "jepa dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0 && d.hidden % 4 == 0, EP_E_SHAPE,
             "jepa: D %% H == 0 and D/H, D, hidden multiples of 4 (D=%d H=%d hidden=%d)", d.D, d.H, d.hidden);
  EP_REQUIRE(d.H <= 32 && (size_t)(2 * d.D + 256) * 4 <= 60000, EP_E_UNSUPPORTED, "jepa: heads > 32 or D too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "jepa head: C must be positive");
  return 0;
}

static int jepa_params_ok(const ep_jepa_params* p, const char* what) {
  EP_REQUIRE(p, EP_E_ARG, "%s: null parameter struct", what);
  const float* ts[] = {p->query, p->n1_w, p->n1_b, p->q_w, p->q_b, p->kv_w, p->kv_b, p->proj_w, p->proj_b, p->n2_w, p->n2_b,
                       p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b};
  for (const float* t : ts) EP_REQUIRE(t && aligned16(t), EP_E_ALIGN, "%s: tensors must be non-null and 16-byte aligned", what);
  return 0;
}

static PoolParams jepa_pool_params(const ep_jepa_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                                   const float* tokstat, const JepaWs& w) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);
  p.cls = w.wq; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index; p.tokstat = tokstat;
  return p;
}

static int jepa_forward_core(const ep_jepa_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const float* tokstat, float ln_eps, const ep_jepa_params& pr, const JepaWs& w, float* out,
                             hipStream_t st) {
  const int D = d.D, dh = D / d.H, Hd = d.hidden, B = d.B;
  const float scale = (float)pow((double)dh, -0.5);                        // modules.py:134 (SDPA default scale)
  if (!tokstat) {
    EP_REQUIRE(!index, EP_E_ARG, "jepa: an indexed token store needs precomputed token statistics");
    EP_TRY(token_stats(x, x_dtype == EP_DTYPE_BF16, bstride, B, d.N, D, ln_eps, w.tstat, st));
    tokstat = w.tstat;
  }
  hipLaunchKernelGGL(ep_siglip_q_kernel, dim3((D + 3) / 4), dim3(256), 0, st, pr.query, pr.q_w, pr.q_b, D, w.q);
  EP_TRY(siglip_u(w.q, pr.kv_w, D, d.H, dh, scale, w.u, st));
  hipLaunchKernelGGL(ep_rowscale_kernel, dim3((d.H * D + 255) / 256), dim3(256), 0, st, w.u, pr.n1_w, d.H, D, w.wq);
  hipLaunchKernelGGL(ep_cae_wv_kernel, dim3((D + 3) / 4), dim3(256), 0, st, pr.kv_w + (int64_t)D * D, pr.n1_w, pr.n1_b, D, w.Wvs,
                     w.bo, pr.kv_b + D);                                    // Wv diag(g1);  Wv b1 + bv
  hipLaunchKernelGGL(ep_vecadd_kernel, dim3((D + 255) / 256), dim3(256), 0, st, pr.proj_b, pr.query, D, w.bq0);   // bp + q0
  EP_LAUNCH_CHECK("ep_jepa query kernels");
  EP_TRY(pool_forward(jepa_pool_params(d, x, x_dtype, bstride, index, tokstat, w), st));
  {
    GemmParams g = mkg(w.P, (int64_t)d.H * D, w.Wvs, D, w.ya, D, B, dh, D);
    g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh; g.bias = w.bo; g.sBiasz = dh;
    EP_TRY(gemm(true, true, g, d.H, st));
  }
  { GemmParams g = mkg(w.ya, D, pr.proj_w, D, w.q1, D, B, D, D); g.bias = w.bq0; EP_TRY(gemm(true, true, g, 1, st)); }   // q1 = q0 + proj(o)
  EP_TRY(token_stats(w.q1, 0, D, B, 1, D, ln_eps, w.qstat, st));                                                        // LN2 statistics
  const int64_t nd = (int64_t)B * D;
  hipLaunchKernelGGL(ep_rowln_apply_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, w.q1, w.qstat, pr.n2_w, pr.n2_b, nd,
                     D, w.h2);
  { GemmParams g = mkg(w.h2, D, pr.fc1_w, D, w.pre, Hd, B, Hd, D); g.bias = pr.fc1_b; EP_TRY(gemm(true, true, g, 1, st)); }
  const int64_t n4 = (int64_t)B * Hd / 4;
  hipLaunchKernelGGL(ep_gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.h1);
  EP_LAUNCH_CHECK("ep_jepa forward kernels");
  EP_HIP(hipMemcpyAsync(out, w.q1, (size_t)nd * sizeof(float), hipMemcpyDeviceToDevice, st));
  { GemmParams g = mkg(w.h1, Hd, pr.fc2_w, Hd, out, D, B, D, Hd); g.bias = pr.fc2_b; g.accumulate = 1;
    g.skws = w.skws; g.skws_floats = w.skws_floats; EP_TRY(gemm(true, true, g, 1, st)); }
  return 0;
}

static int jepa_backward_core(const ep_jepa_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                              const float* tokstat, const ep_jepa_params& pr, const float* dout, const ep_jepa_params& gr, int acc,
                              const JepaWs& w, SideTasks sd, hipStream_t st, hipStream_t aux) {
  const int D = d.D, dh = D / d.H, Hd = d.hidden, B = d.B;
  const float scale = (float)pow((double)dh, -0.5);
  const int64_t n4 = (int64_t)B * Hd / 4;
  if (!tokstat) tokstat = w.tstat;
  // The weight-gradient contractions feed nothing before the optimizer (dWv: the small kernels behind the second pass); the
  // token-pass kernel of this head takes no side workgroups: the aux stream, each contraction as early as its operands exist
  // (AuxSide, ep_internal.h).  At 256 x 768, 1024 images: 1.367 -> 1.27 ms per step.  Round 6: successive forks alternate between the
  // aux stream and the library's second side queue -- on one queue the five contractions (45 - 80 us each) ran into the second pass,
  // which starved the last one (236 us) and the optimizer waited ~95 us for it: 1.266 -> 1.209 ms (EP_AUX_TWO=0: one queue).
  AuxSide ax;
  EP_TRY(aux_side_begin(ax, st, aux, true));
  GemmParams gW2 = mkg(dout, D, w.h1, Hd, gr.fc2_w, Hd, D, Hd, B); gW2.accumulate = acc; gW2.side = 1;
  GemmParams gW1 = mkg(w.dh1, Hd, w.h2, D, gr.fc1_w, D, Hd, D, B); gW1.accumulate = acc; gW1.side = 1;
  GemmParams gWp = mkg(w.dq1, D, w.ya, D, gr.proj_w, D, D, D, B); gWp.accumulate = acc; gWp.side = 1;
  GemmParams gWv = mkg(w.dya, D, w.P, (int64_t)d.H * D, w.dWvs, D, dh, D, B);
  gWv.sAz = dh; gWv.extA = dh; gWv.sBz = D; gWv.extB = D; gWv.sCz = (int64_t)dh * D; gWv.side = 1;
  EP_REQUIRE(gemm_side_ok(gW2, false, false) && gemm_side_ok(gW1, false, false) && gemm_side_ok(gWp, false, false) &&
             gemm_side_ok(gWv, false, false), EP_E_ALIGN, "jepa: unaligned gradient contraction");
  side_add_gemm(sd, gW2, 1);
  EP_TRY(aux_side_fork(ax, sd));                     // the caller's dWc, dW2 = dout^T h1
  // out = q1 + fc2(gelu(fc1(LN2(q1))))
  EP_TRY(gemm(true, false, mkg(dout, D, pr.fc2_w, Hd, w.dh1, Hd, B, Hd, D), 1, st));               // dh1 = dout W2
  hipLaunchKernelGGL(ep_gelu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.dh1);   // -> dpre
  EP_LAUNCH_CHECK("ep_gelu_bwd_kernel");
  side_add_gemm(sd, gW1, 1);
  EP_TRY(aux_side_fork(ax, sd));                     // dW1 = dpre^T h2
  { GemmParams g = mkg(w.dh1, Hd, pr.fc1_w, D, w.dh2, D, B, D, Hd); g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(true, false, g, 1, st)); }                                                          // dh2 = dpre W1
  hipLaunchKernelGGL(ep_rowln_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, w.dh2, w.q1, w.qstat, pr.n2_w, dout, B, D, w.dq1);
  side_add_gemm(sd, gWp, 1);
  EP_TRY(aux_side_fork(ax, sd));                     // dWp = dq1^T ya
  EP_TRY(lnaffine_grad(w.dh2, w.q1, w.qstat, B, D, acc, gr.n2_w,
                     gr.n2_b, st));
  EP_LAUNCH_CHECK("ep_jepa LN2 backward kernels");
  if (!side_add_colsum(sd, dout, B, D, D, acc, gr.fc2_b)) EP_TRY(colsum(dout, B, D, D, acc, gr.fc2_b, st));
  if (!side_add_colsum(sd, w.dh1, B, Hd, Hd, acc, gr.fc1_b)) EP_TRY(colsum(w.dh1, B, Hd, Hd, acc, gr.fc1_b, st));
  if (!side_add_colsum(sd, w.dq1, B, D, D, acc, gr.proj_b)) EP_TRY(colsum(w.dq1, B, D, D, acc, gr.proj_b, st));   // q1 = q0 + o Wp^T + bp
  EP_TRY(gemm(true, false, mkg(w.dq1, D, pr.proj_w, D, w.dya, D, B, D, D), 1, st));                 // dya = dq1 Wp
  side_add_gemm(sd, gWv, d.H);
  EP_TRY(aux_side_fork(ax, sd));                     // dWv = dya^T P
  EP_TRY(aux_side_rest(ax, sd));                     // (beside the pass one of these 8-us column sums took the whole pass: 200 us)
  EP_TRY(colsum(w.dya, B, D, D, 0, w.dbo, st));                                                     // d(Wv b1 + bv)
  EP_TRY(delta_rows(w.dya, w.ya, B * d.H, dh, w.ML, st, w.bo, d.H));
  {
    GemmParams g = mkg(w.dya, D, w.Wvs, D, w.dP, (int64_t)d.H * D, B, D, dh);
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D;
    EP_TRY(gemm(true, false, g, d.H, st));
  }
  PoolParams p = jepa_pool_params(d, x, x_dtype, bstride, index, tokstat, w);
  p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
  EP_TRY(aux_side_before_pass(ax, sd));
  EP_TRY(pool_backward(p, w.dw, 0, st));
  EP_TRY(aux_side_join(ax));
  // value side: d kv.weight[D:], d kv.bias[D:], and the value-side parts of d norm1.weight / bias
  hipLaunchKernelGGL(ep_cae_dwv_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.dWvs, w.dbo, pr.kv_w + (int64_t)D * D, pr.n1_w,
                     pr.n1_b, D, acc, gr.kv_w + (int64_t)D * D, gr.n1_w, gr.n1_b, (float*)nullptr, (float*)nullptr);
  EP_TRY(colsum(w.dya, B, D, D, acc, gr.kv_b + D, st));
  // key side: du = g1 * dw;  d norm1.weight += sum_h u_h * dw_h   (accumulating onto the value-side part)
  hipLaunchKernelGGL(ep_cae_du_kernel, dim3((D + 255) / 256), dim3(256), 0, st, w.dw, w.u, pr.n1_w, D, d.H, 1, w.du, gr.n1_w,
                     gr.n1_b);
  // query chain (as in the SigLIP head): d q.weight / bias, d kv.weight[:D], d kv.bias[:D] = 0, d query += Wq^T dq
  hipLaunchKernelGGL(ep_siglip_dq_kernel, dim3((D + 3) / 4), dim3(256), 0, st, w.du, pr.kv_w, D, dh, scale, acc, w.dq, gr.q_b);
  EP_TRY(siglip_qgrad(w.q, w.dq, w.du,
                     pr.query, pr.q_w, D, dh, scale, acc, gr.kv_w, gr.q_w, gr.query, gr.kv_b, st));
  EP_LAUNCH_CHECK("ep_jepa backward kernels");
  EP_TRY(colsum(w.dq1, B, D, D, 1, gr.query, st));             // + the direct path q1 = q0 + ... (added onto Wq^T dq)
  return 0;
}

static ep_jepa_params jepa_views(float* base, const int64_t o[JEPA_NT]) {
  ep_jepa_params p;
  p.query = base + o[0]; p.n1_w = base + o[1]; p.n1_b = base + o[2]; p.q_w = base + o[3]; p.q_b = base + o[4];
  p.kv_w = base + o[5]; p.kv_b = base + o[6]; p.proj_w = base + o[7]; p.proj_b = base + o[8]; p.n2_w = base + o[9];
  p.n2_b = base + o[10]; p.fc1_w = base + o[11]; p.fc1_b = base + o[12]; p.fc2_w = base + o[13]; p.fc2_b = base + o[14];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

size_t ep_siglip_pool_workspace_bytes(const ep_siglip_dims* dims) {
  if (!dims || sig_check(*dims, false) != 0) return 0;
  return sig_carve(*dims, nullptr, false).total;
}

int ep_siglip_pool_forward(const ep_siglip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                           const int32_t* image_index, const ep_siglip_params* params, float* out, void* ws,
                           size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && out && ws, EP_E_ARG, "ep_siglip_pool_forward: null pointer");
  EP_TRY(sig_check(*dims, false));
  EP_TRY(sig_params_ok(params, "ep_siglip_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(out), EP_E_ALIGN, "ep_siglip_pool_forward: out / ws must be 16-byte aligned");
  const SigWs w = sig_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_siglip_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return sig_forward_core(*dims, x, x_dtype, x_bstride, image_index, *params, w, out, (hipStream_t)stream);
}

int ep_siglip_pool_backward(const ep_siglip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const int32_t* image_index, const ep_siglip_params* params, const float* dout,
                            const ep_siglip_params* grads, int accumulate, void* ws, size_t ws_bytes,
                            ep_stream_t stream) {
  EP_REQUIRE(dims && dout && ws, EP_E_ARG, "ep_siglip_pool_backward: null pointer");
  EP_TRY(sig_check(*dims, false));
  EP_TRY(sig_params_ok(params, "ep_siglip_pool_backward(params)"));
  EP_TRY(sig_params_ok(grads, "ep_siglip_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dout), EP_E_ALIGN, "ep_siglip_pool_backward: dout / ws must be 16-byte aligned");
  const SigWs w = sig_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_siglip_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return sig_backward_core(*dims, x, x_dtype, x_bstride, image_index, *params, dout, *grads, accumulate, w, SideTasks{},
                           (hipStream_t)stream, nullptr);
}

int ep_siglip_attention(const ep_siglip_dims* dims, const void* ws, float* A, ep_stream_t stream) {
  EP_REQUIRE(dims && ws && A, EP_E_ARG, "ep_siglip_attention: null pointer");
  EP_TRY(sig_check(*dims, false));
  const SigWs w = sig_carve(*dims, const_cast<void*>(ws), false);
  return attention_from_scores(w.S, w.ML, dims->B * dims->H, dims->N, A, (hipStream_t)stream);
}

int64_t ep_siglip_head_param_offsets(const ep_siglip_dims* dims, int64_t offsets[13]) { return sig_offsets(*dims, offsets); }

size_t ep_siglip_head_workspace_bytes(const ep_siglip_dims* dims) {
  if (!dims || sig_check(*dims, true) != 0) return 0;
  return sig_carve(*dims, nullptr, true).total;
}

int ep_siglip_head_train_step(const ep_siglip_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_siglip_head_train_step: null pointer");
  const ep_siglip_dims& d = s->dims;
  EP_TRY(sig_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const SigWs w = sig_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_siglip_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[SIG_NT];
  const int64_t total = sig_offsets(d, offs);
  const ep_siglip_params pr = sig_views(s->params, offs), gr = sig_views(s->grads, offs);
  float* Wc = s->params + offs[11]; float* bc = s->params + offs[12];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    EP_TRY(sig_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, nullptr, nullptr, 0, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    SideTasks sd{};
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, d.D, d.C, s->grads + offs[11], s->accumulate);
    EP_REQUIRE(gemm_side_ok(gWc, false, false), EP_E_ALIGN, "siglip head: unaligned classifier gradient");
    side_add_gemm(sd, gWc, 1);
    sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[12]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
    sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
    sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
    sd.total += sd.n_colsum + sd.n_stats;
    EP_TRY(sig_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, pr, w.dy, gr, s->accumulate, w, sd, st,
                             (hipStream_t)s->aux_stream));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t D = d.D, Hd = d.hidden;
    const int64_t sizes[SIG_NT] = {D, D * D, D, 2 * D * D, 2 * D, D * D, D, Hd * D, Hd, D * Hd, D, (int64_t)d.C * D, d.C};
    // util/lars.py:22: trust ratio + weight decay for tensors with ndim > 1; the latent is (1, 1, D)
    const int trust[SIG_NT] = {1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0};
    ep_segment segs[SIG_NT];
    for (int i = 0; i < SIG_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? SIG_NT : 0, s->lr, s->weight_decay,
                      s->momentum, s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step,
                      s->found_inf, s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_siglip_head_eval_forward(const ep_siglip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                                const int32_t* image_index, const float* params, const float* running_mean,
                                const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                                ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_siglip_head_eval_forward: null pointer");
  const ep_siglip_dims& d = *dims;
  EP_TRY(sig_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const SigWs w = sig_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_siglip_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[SIG_NT];
  sig_offsets(d, offs);
  const ep_siglip_params pr = sig_views(const_cast<float*>(params), offs);
  EP_TRY(sig_forward_core(d, x, x_dtype, x_bstride, image_index, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[11], params + offs[12], d.B, d.D, d.C, logits, ldl, st);
}


/* ---- V-JEPA attentive pooler ---------------------------------------------------------------------------- */
size_t ep_jepa_pool_workspace_bytes(const ep_jepa_dims* dims) {
  if (!dims || jepa_check(*dims, false) != 0) return 0;
  return jepa_carve(*dims, nullptr, false).total;
}

int ep_jepa_pool_forward(const ep_jepa_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                         const float* token_stats_, float ln_eps, const ep_jepa_params* params, float* out, void* ws,
                         size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && out && ws, EP_E_ARG, "ep_jepa_pool_forward: null pointer");
  EP_TRY(jepa_check(*dims, false));
  EP_TRY(jepa_params_ok(params, "ep_jepa_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(out), EP_E_ALIGN, "ep_jepa_pool_forward: out / ws must be 16-byte aligned");
  const JepaWs w = jepa_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_jepa_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return jepa_forward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, ln_eps, *params, w, out, (hipStream_t)stream);
}

int ep_jepa_pool_backward(const ep_jepa_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                          const float* token_stats_, const ep_jepa_params* params, const float* dout,
                          const ep_jepa_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dout && ws, EP_E_ARG, "ep_jepa_pool_backward: null pointer");
  EP_TRY(jepa_check(*dims, false));
  EP_TRY(jepa_params_ok(params, "ep_jepa_pool_backward(params)"));
  EP_TRY(jepa_params_ok(grads, "ep_jepa_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dout), EP_E_ALIGN, "ep_jepa_pool_backward: dout / ws must be 16-byte aligned");
  const JepaWs w = jepa_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_jepa_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return jepa_backward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, *params, dout, *grads, accumulate, w,
                            SideTasks{}, (hipStream_t)stream, nullptr);
}

int64_t ep_jepa_head_param_offsets(const ep_jepa_dims* dims, int64_t offsets[17]) { return jepa_offsets(*dims, offsets); }

size_t ep_jepa_head_workspace_bytes(const ep_jepa_dims* dims) {
  if (!dims || jepa_check(*dims, true) != 0) return 0;
  return jepa_carve(*dims, nullptr, true).total;
}

int ep_jepa_head_train_step(const ep_jepa_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_jepa_head_train_step: null pointer");
  const ep_jepa_dims& d = s->dims;
  EP_TRY(jepa_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const JepaWs w = jepa_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_jepa_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[JEPA_NT];
  const int64_t total = jepa_offsets(d, offs);
  const ep_jepa_params pr = jepa_views(s->params, offs), gr = jepa_views(s->grads, offs);
  float* Wc = s->params + offs[15]; float* bc = s->params + offs[16];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    EP_TRY(jepa_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, s->ln_eps, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, nullptr, nullptr, 0, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    SideTasks sd{};
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, d.D, d.C, s->grads + offs[15], s->accumulate);
    EP_REQUIRE(gemm_side_ok(gWc, false, false), EP_E_ALIGN, "jepa head: unaligned classifier gradient");
    side_add_gemm(sd, gWc, 1);
    sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[16]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
    sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
    sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
    sd.total += sd.n_colsum + sd.n_stats;
    EP_TRY(jepa_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, pr, w.dy, gr, s->accumulate, w, sd,
                              st, (hipStream_t)s->aux_stream));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t D = d.D, Hd = d.hidden;
    const int64_t sizes[JEPA_NT] = {D, D, D, D * D, D, 2 * D * D, 2 * D, D * D, D, D, D, Hd * D, Hd, D * Hd, D, (int64_t)d.C * D, d.C};
    const int trust[JEPA_NT] = {1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0};     // ndim > 1 (util/lars.py:22); query is (1,1,D)
    ep_segment segs[JEPA_NT];
    for (int i = 0; i < JEPA_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? JEPA_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_jepa_head_eval_forward(const ep_jepa_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                              const float* token_stats_, float ln_eps, const float* params, const float* running_mean,
                              const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                              ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_jepa_head_eval_forward: null pointer");
  const ep_jepa_dims& d = *dims;
  EP_TRY(jepa_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const JepaWs w = jepa_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_jepa_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[JEPA_NT];
  jepa_offsets(d, offs);
  const ep_jepa_params pr = jepa_views(const_cast<float*>(params), offs);
  EP_TRY(jepa_forward_core(d, x, x_dtype, x_bstride, image_index, token_stats_, ln_eps, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[15], params + offs[16], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
