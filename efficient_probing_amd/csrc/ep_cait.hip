// CaiT class-attention pooling (reference poolings/other_pool.py:390-507: CAPooling with one LayerScale_Block_CA /
// Class_Attention; registry entry probe_heads.py:79: CAPooling(embed_dim=dim) -> 4 heads, qkv bias, LayerNorm eps 1e-6,
// LayerScale init 1e-5, MLP x4, final LayerNorm eps 1e-5).  With c = cls_token:
//     u   = norm1([c ; x])                              (N + 1 rows: the class row is batch independent)
//     a   = proj(softmax(scale q(u_0) k(u)^T) v(u))     (only the class row queries; it is also a key / value)
//     c1  = c + gamma_1 * a ;  c2 = c1 + gamma_2 * mlp(norm2(c1)) ;  out = norm(c2)
// The patch rows are the LayerNorm-of-tokens mode of the EP passes with the derived query rows w_h = g1 * (scale Wk_h^T q_h)
// (all additive constants -- key bias, Wk b1 -- shift the N + 1 scores of a head equally and cancel).  The class row is ONE
// extra softmax entry per head with score s_c[h] = w_h . chat (chat = normalised cls_token) and value Wv'_h chat: it is
// merged into the pass's online-softmax state after the pass,
//     m' = max(m, s_c) ; l' = l e^(m-m') + e^(s_c-m') ; rho = l e^(m-m') / l' ; a_c = 1 - rho
//     o[b,h] = rho (Wv'_h Phat[b,h]) + a_c (Wv'_h chat) + (Wv b1 + bv)_h ,     Wv' = Wv diag(g1)
// and the second pass simply runs with (m', l') and delta' = dO . (o - bias): it then produces the softmax gradient of the
// (N + 1)-entry distribution for the patch rows; the class entry's own gradient is a (B, H) computation.
#include "ep_side.h"
#include "ep_lnaffine.h"
#include "ep_headkernels.h"

namespace ep {

// LayerNorm of the class token: chat = (c - mean) rstd, un0 = g1 * chat + b1 ; lnstat = {mean, rstd}   (one workgroup)
__global__ __launch_bounds__(256) void ep_cait_cls_kernel(const float* __restrict__ c, const float* __restrict__ g1,
                                                        const float* __restrict__ b1, int D, float eps,
                                                        float* __restrict__ chat, float* __restrict__ un0,
                                                        float* __restrict__ lnstat) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s += c[d];
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
  __syncthreads();
  float q = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) { const float e = c[d] - mean; q = fmaf(e, e, q); }
  q = wave_sum(q);
  if (lane == 0) red[w] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)D + eps);
  for (int d = threadIdx.x; d < D; d += 256) {
    const float ch = (c[d] - mean) * rstd;
    chat[d] = ch;
    un0[d] = fmaf(g1[d], ch, b1[d]);
  }
  if (threadIdx.x == 0) { lnstat[0] = mean; lnstat[1] = rstd; }
}

// rows 0 .. D-1: vc[j] = Wvs[j,:] . chat ; rows D .. D+H-1: sc[h] = wq[h,:] . chat        (one wave per row)
__global__ __launch_bounds__(256) void ep_cait_clsrow_kernel(const float* __restrict__ Wvs, const float* __restrict__ wq,
                                                           const float* __restrict__ chat, int D, int H,
                                                           float* __restrict__ vc, float* __restrict__ sc) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D + H) return;
  const int lane = threadIdx.x & 63;
  const float* row = j < D ? Wvs + (int64_t)j * D : wq + (int64_t)(j - D) * D;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(row[d], chat[d], acc);
  acc = wave_sum(acc);
  if (lane == 0) { if (j < D) vc[j] = acc; else sc[j - D] = acc; }
}

// merge the class entry into every (image, head) softmax:  ya = rho ya0 + a_c vc + bo ;  ML2 = {m', l', -, -} ; mix = {rho, a_c}
__global__ __launch_bounds__(256) void ep_cait_merge_kernel(const float* __restrict__ ya0, const float* __restrict__ ML,
                                                          const float* __restrict__ sc, const float* __restrict__ vc,
                                                          const float* __restrict__ bo, int64_t total, int D, int dh, int H,
                                                          float* __restrict__ ya, float* __restrict__ ML2,
                                                          float* __restrict__ mix) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / D; const int j = (int)(i % D), h = j / dh;
  const float m = ML[(b * H + h) * 4], l = ML[(b * H + h) * 4 + 1], s = sc[h];
  const float mn = fmaxf(m, s);
  const float et = l * expf(m - mn), ec = expf(s - mn);
  const float ln = et + ec, rho = et / ln, ac = ec / ln;
  ya[i] = fmaf(rho, ya0[i], fmaf(ac, vc[j], bo[j]));
  if (j % dh == 0) {
    ML2[(b * H + h) * 4] = mn; ML2[(b * H + h) * 4 + 1] = ln;
    mix[(b * H + h) * 2] = rho; mix[(b * H + h) * 2 + 1] = ac;
  }
}

// out[b,:] = a[(b),:] + gamma * z[b,:]      (a_bstride = 0: the same row for every image)
__global__ __launch_bounds__(256) void ep_cait_res_kernel(const float* __restrict__ a, int64_t a_bstride,
                                                        const float* __restrict__ gamma, const float* __restrict__ z,
                                                        int64_t total, int D, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  out[i] = fmaf(gamma[d], z[i], a[(i / D) * a_bstride + d]);
}

// dz = dc * gamma (element-wise) ; d gamma[d] (+)= sum_b dc[b,d] z[b,d] ; optionally dsum[d] = sum_b dc[b,d]
__global__ __launch_bounds__(256) void ep_cait_scale_bwd_kernel(const float* __restrict__ dc, const float* __restrict__ z,
                                                              const float* __restrict__ gamma, int B, int D, int accumulate,
                                                              float* __restrict__ dz, float* __restrict__ dgamma,
                                                              float* __restrict__ dsum) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * CG + tx;
  const bool ok = c < D;
  float sg = 0.f, ss = 0.f;
  if (ok) {
    const float g = gamma[c];
    for (int b = ty; b < B; b += RL) {
      const int64_t i = (int64_t)b * D + c;
      const float v = dc[i];
      dz[i] = v * g;
      sg = fmaf(v, z[i], sg);
      ss += v;
    }
  }
  sg = colreduce(sg, sm, tx, ty);
  ss = colreduce(ss, sm, tx, ty);
  if (ty == 0 && ok) {
    dgamma[c] = accumulate ? dgamma[c] + sg : sg;
    if (dsum) dsum[c] = ss;                              // a temporary of this step, never accumulated
  }
}

// class-entry part of the attention backward, per (image, head) -- one wave each:
//   dO = dya[b, slice h] ; dA_c = dO . vc[slice h] ; dS_c = a_c (dA_c - delta') ; dya0 = rho dya (in place into dya0)
//   csc[b,h] = {dS_c, a_c}
__global__ __launch_bounds__(256) void ep_cait_clsgrad_kernel(const float* __restrict__ dya, const float* __restrict__ vc,
                                                            const float* __restrict__ ML2, const float* __restrict__ mix,
                                                            int rows, int dh, int H, float* __restrict__ dya0,
                                                            float* __restrict__ csc) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63, h = r % H;
  const float rho = mix[r * 2], ac = mix[r * 2 + 1];
  float s = 0.f;
  for (int c = lane; c < dh; c += 64) {
    const float g = dya[(int64_t)r * dh + c];
    s = fmaf(g, vc[h * dh + c], s);
    dya0[(int64_t)r * dh + c] = rho * g;
  }
  s = wave_sum(s);
  if (lane == 0) { csc[r * 2] = ac * (s - ML2[(int64_t)r * 4 + 2]); csc[r * 2 + 1] = ac; }
}

// batch reductions of the class entry, per 16 columns d:
//   dw[h,d]   += chat[d] sum_b dS_c[b,h]
//   dchat[d]   = sum_h wq[h,d] sum_b dS_c[b,h] + sum_b sum_h a_c[b,h] dP[b,h,d]
// (two launches: ep_cait_clsred_part_kernel sums dS_c[r] dP[r,:] over a sixteenth of the B H rows per workgroup row --
// 1024 threads = 32 column lanes x 32 row lanes -- into part[chunk][D]; ep_cait_clsred_kernel adds the chunks in order and
// finishes the class-row terms)
constexpr int CAIT_RS = 16;
__global__ __launch_bounds__(1024) void ep_cait_clsred_part_kernel(const float* __restrict__ csc, const float* __restrict__ dP,
                                                                 int64_t rows, int D, float* __restrict__ part) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float acc = 0.f;
  if (d < D)
    for (int64_t r = r0 + ty; r < r1; r += 32) acc = fmaf(csc[r * 2 + 1], dP[r * D + d], acc);
  sm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && d < D) {
    float g = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) g += sm[i][tx];
    part[(int64_t)blockIdx.y * D + d] = g;
  }
}
__global__ __launch_bounds__(1024) void ep_cait_clsred_kernel(const float* __restrict__ csc, const float* __restrict__ part, int nparts,
                                                            const float* __restrict__ wq, const float* __restrict__ chat, int B,
                                                            int H, int D, float* __restrict__ dw, float* __restrict__ dchat) {
  __shared__ float sm[32][33];
  __shared__ float sds[32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx;
  const bool ok = d < D;
  // sum_b dS_c[b,h] for every head (H <= 32): row lane ty sums b = ty, ty + 32, ..., the 32 partials in a fixed order
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    if (tx == 0) for (int b = ty; b < B; b += 32) s += csc[((int64_t)b * H + h) * 2];
    __syncthreads();
    if (tx == 0) sm[ty][0] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < 32; ++i) t += sm[i][0];
      sds[h] = t;
    }
  }
  __syncthreads();
  if (ty == 0 && ok) {
    float g = 0.f;
    for (int i = 0; i < nparts; ++i) g += part[(int64_t)i * D + d];
    const float ch = chat[d];
    for (int h = 0; h < H; ++h) {
      g = fmaf(wq[(int64_t)h * D + d], sds[h], g);
      dw[(int64_t)h * D + d] += ch * sds[h];
    }
    dchat[d] = g;
  }
}

// dvc[j] = sum_b a_c[b,h(j)] dya[b,j]       (one 16-column block per workgroup)
__global__ __launch_bounds__(256) void ep_cait_dvc_kernel(const float* __restrict__ dya, const float* __restrict__ mix, int B,
                                                        int D, int dh, int H, float* __restrict__ dvc) {
  __shared__ float sm[RL][CG];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int j = blockIdx.x * CG + tx;
  const bool ok = j < D;
  float s = 0.f;
  if (ok) {
    const int h = j / dh;
    for (int b = ty; b < B; b += RL) s = fmaf(mix[((int64_t)b * H + h) * 2 + 1], dya[(int64_t)b * D + j], s);
  }
  s = colreduce(s, sm, tx, ty);
  if (ty == 0 && ok) dvc[j] = s;
}

// dWvs[j,d] += dvc[j] chat[d]
__global__ __launch_bounds__(256) void ep_cait_rank1_kernel(const float* __restrict__ dvc, const float* __restrict__ chat, int D,
                                                          float* __restrict__ dWvs) {
  const int d = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (d < D) dWvs[(int64_t)j * D + d] += dvc[j] * chat[d];
}

// LayerNorm backward of the class row with both kinds of incoming gradient:
//   dun0 (w.r.t. the affine output g1 * chat + b1) and dchat (w.r.t. the normalised row)
//   d g1 += dun0 * chat ; d b1 += dun0 ; dch = dun0 * g1 + dchat ; d c (+)= rstd (dch - mean(dch) - chat mean(dch chat)) + dcsum
__global__ __launch_bounds__(256) void ep_cait_clsln_bwd_kernel(const float* __restrict__ dun0, const float* __restrict__ dchat,
                                                              const float* __restrict__ chat, const float* __restrict__ g1,
                                                              const float* __restrict__ lnstat, const float* __restrict__ dcsum,
                                                              int D, int accumulate, float* __restrict__ dg1,
                                                              float* __restrict__ db1, float* __restrict__ dc) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float dch = fmaf(dun0[d], g1[d], dchat[d]);
    s1 += dch; s2 = fmaf(dch, chat[d], s2);
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[0][w] = s1; red[1][w] = s2; }
  __syncthreads();
  const float m1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
  const float m2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
  const float rstd = lnstat[1];
  for (int d = threadIdx.x; d < D; d += 256) {
    const float du_ = dun0[d], ch = chat[d];
    const float dch = fmaf(du_, g1[d], dchat[d]);
    dg1[d] += du_ * ch;                                 // (accumulates onto the value / key side parts written before)
    db1[d] += du_;
    const float v = rstd * (dch - m1 - ch * m2) + dcsum[d];
    dc[d] = accumulate ? dc[d] + v : v;
  }
}

// ---------------------------------------------------------------------------------------------
constexpr int CAIT_NT = 23;   // cls | gamma_1 gamma_2 | n1.w n1.b | q.w q.b k.w k.b v.w v.b | proj.w proj.b | n2.w n2.b |
                              // fc1.w fc1.b fc2.w fc2.b | norm.w norm.b | fc.weight fc.bias
struct CaitWs {
  float *P, *S, *ML, *ML2, *mix, *csc, *tstat, *ya0, *ya, *z1, *c1, *stat2, *h2, *pre, *h1, *m2, *c2, *statf;
  float *dc2, *dm2, *dh1, *dh2, *dc1, *dz1, *dya, *dya0, *dP;
  float *chat, *un0, *lnstat, *q, *u, *wq, *sc, *vc, *dvc, *dw, *du, *dq, *dun0, *dchat, *dcsum, *Wvs, *bo, *dWvs, *dbo, *scr, *cpart;
  float* skws; size_t skws_floats;                   // K-slice scratch of the two long-K MLP contractions (ep_gemm.hip: gemm_split_k)
  void* pool_ws; size_t pool_ws_bytes;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static void cait_sizes(const ep_cait_dims& d, int64_t sizes[CAIT_NT]) {
  const int64_t D = d.D, Hd = d.hidden;
  const int64_t s[CAIT_NT] = {D, D, D, D, D, D * D, D, D * D, D, D * D, D, D * D, D, D, D, Hd * D, Hd, D * Hd, D, D, D,
                              (int64_t)d.C * D, d.C};
  for (int i = 0; i < CAIT_NT; ++i) sizes[i] = s[i];
}
static int64_t cait_offsets(const ep_cait_dims& d, int64_t offs[CAIT_NT]) {
  int64_t sizes[CAIT_NT];
  cait_sizes(d, sizes);
  int64_t off = 0;
  for (int i = 0; i < CAIT_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static CaitWs cait_carve(const ep_cait_dims& d, void* base, bool head) {
  CaitWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, Hd = d.hidden, H = d.H;
  w.P = take(B * H * D); w.S = take(B * H * d.N); w.ML = take(B * H * 4); w.ML2 = take(B * H * 4); w.mix = take(B * H * 2);
  w.csc = take(B * H * 2); w.tstat = take(B * d.N * 2);
  w.ya0 = take(B * D); w.ya = take(B * D); w.z1 = take(B * D); w.c1 = take(B * D); w.stat2 = take(B * 2); w.h2 = take(B * D);
  w.pre = take(B * Hd); w.h1 = take(B * Hd); w.m2 = take(B * D); w.c2 = take(B * D); w.statf = take(B * 2);
  w.dc2 = take(B * D); w.dm2 = take(B * D); w.dh1 = take(B * Hd); w.dh2 = take(B * D); w.dc1 = take(B * D); w.dz1 = take(B * D);
  w.dya = take(B * D); w.dya0 = take(B * D); w.dP = take(B * H * D);
  w.chat = take(D); w.un0 = take(D); w.lnstat = take(4); w.q = take(D); w.u = take(H * D); w.wq = take(H * D); w.sc = take(32);
  w.vc = take(D); w.dvc = take(D); w.dw = take(H * D); w.du = take(H * D); w.dq = take(D); w.dun0 = take(D); w.dchat = take(D); w.cpart = take((size_t)CAIT_RS * D);
  w.skws_floats = (size_t)4 * d.B * D; w.skws = take(w.skws_floats);
  w.dcsum = take(D); w.Wvs = take(D * D); w.bo = take(D); w.dWvs = take(D * D); w.dbo = take(D); w.scr = take(2 * D);
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.H);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[CAIT_NT];
    w.opt_ws_bytes = optim_workspace_bytes(cait_offsets(d, offs), CAIT_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int cait_check(const ep_cait_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0 && d.hidden > 0, EP_E_ARG, "cait dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0 && d.hidden % 4 == 0, EP_E_SHAPE,
             "cait: D %% H == 0 and D/H, D, hidden multiples of 4 (D=%d H=%d hidden=%d)", d.D, d.H, d.hidden);
  EP_REQUIRE(d.H <= 32 && (size_t)(2 * d.D + 256) * 4 <= 60000, EP_E_UNSUPPORTED, "cait: heads > 32 or D too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "cait head: C must be positive");
  return 0;
}

static int cait_params_ok(const ep_cait_params* p, const char* what) {
  EP_REQUIRE(p, EP_E_ARG, "%s: null parameter struct", what);
  const float* ts[] = {p->cls_token, p->gamma_1, p->gamma_2, p->n1_w, p->n1_b, p->q_w, p->q_b, p->k_w, p->k_b, p->v_w, p->v_b,
                       p->proj_w, p->proj_b, p->n2_w, p->n2_b, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, p->norm_w, p->norm_b};
  for (const float* t : ts) EP_REQUIRE(t && aligned16(t), EP_E_ALIGN, "%s: tensors must be non-null and 16-byte aligned", what);
  return 0;
}

static GemmParams cgm(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

static PoolParams cait_pool_params(const ep_cait_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                                   const float* tokstat, const CaitWs& w) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);
  p.cls = w.wq; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index; p.tokstat = tokstat;
  return p;
}

static int cait_forward_core(const ep_cait_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const float* tokstat, const ep_cait_params& pr, const CaitWs& w, float* out, hipStream_t st) {
  const int D = d.D, dh = D / d.H, Hd = d.hidden, B = d.B, H = d.H;
  const float scale = (float)pow((double)dh, -0.5);                        // other_pool.py:446-447
  if (!tokstat) {
    EP_REQUIRE(!index, EP_E_ARG, "cait: an indexed token store needs precomputed token statistics");
    EP_TRY(token_stats(x, x_dtype == EP_DTYPE_BF16, bstride, B, d.N, D, d.ln_eps, w.tstat, st));
    tokstat = w.tstat;
  }
  const int64_t nd = (int64_t)B * D;
  const unsigned eg = (unsigned)((nd + 255) / 256);
  hipLaunchKernelGGL(ep_cait_cls_kernel, dim3(1), dim3(256), 0, st, pr.cls_token, pr.n1_w, pr.n1_b, D, d.ln_eps, w.chat, w.un0,
                     w.lnstat);
  hipLaunchKernelGGL(ep_siglip_q_kernel, dim3((D + 3) / 4), dim3(256), 0, st, w.un0, pr.q_w, pr.q_b, D, w.q);
  EP_TRY(siglip_u(w.q, pr.k_w, D, H, dh, scale, w.u, st));
  hipLaunchKernelGGL(ep_rowscale_kernel, dim3((H * D + 255) / 256), dim3(256), 0, st, w.u, pr.n1_w, H, D, w.wq);
  hipLaunchKernelGGL(ep_cae_wv_kernel, dim3((D + 3) / 4), dim3(256), 0, st, pr.v_w, pr.n1_w, pr.n1_b, D, w.Wvs, w.bo, pr.v_b);
  hipLaunchKernelGGL(ep_cait_clsrow_kernel, dim3((D + H + 3) / 4), dim3(256), 0, st, w.Wvs, w.wq, w.chat, D, H, w.vc, w.sc);
  EP_LAUNCH_CHECK("ep_cait query kernels");
  EP_TRY(pool_forward(cait_pool_params(d, x, x_dtype, bstride, index, tokstat, w), st));
  {
    GemmParams g = cgm(w.P, (int64_t)H * D, w.Wvs, D, w.ya0, D, B, dh, D);            // ya0_h = Phat_h Wv'_h^T
    g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh;
    EP_TRY(gemm(true, true, g, H, st));
  }
  hipLaunchKernelGGL(ep_cait_merge_kernel, dim3(eg), dim3(256), 0, st, w.ya0, w.ML, w.sc, w.vc, w.bo, nd, D, dh, H, w.ya, w.ML2,
                     w.mix);
  EP_LAUNCH_CHECK("ep_cait_merge_kernel");
  { GemmParams g = cgm(w.ya, D, pr.proj_w, D, w.z1, D, B, D, D); g.bias = pr.proj_b; EP_TRY(gemm(true, true, g, 1, st)); }
  hipLaunchKernelGGL(ep_cait_res_kernel, dim3(eg), dim3(256), 0, st, pr.cls_token, (int64_t)0, pr.gamma_1, w.z1, nd, D, w.c1);
  EP_TRY(token_stats(w.c1, 0, D, B, 1, D, d.ln_eps, w.stat2, st));
  hipLaunchKernelGGL(ep_rowln_apply_kernel, dim3(eg), dim3(256), 0, st, w.c1, w.stat2, pr.n2_w, pr.n2_b, nd, D, w.h2);
  { GemmParams g = cgm(w.h2, D, pr.fc1_w, D, w.pre, Hd, B, Hd, D); g.bias = pr.fc1_b; EP_TRY(gemm(true, true, g, 1, st)); }
  const int64_t n4 = (int64_t)B * Hd / 4;
  hipLaunchKernelGGL(ep_gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.h1);
  { GemmParams g = cgm(w.h1, Hd, pr.fc2_w, Hd, w.m2, D, B, D, Hd); g.bias = pr.fc2_b; g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(true, true, g, 1, st)); }
  hipLaunchKernelGGL(ep_cait_res_kernel, dim3(eg), dim3(256), 0, st, w.c1, (int64_t)D, pr.gamma_2, w.m2, nd, D, w.c2);
  EP_TRY(token_stats(w.c2, 0, D, B, 1, D, d.final_eps, w.statf, st));
  hipLaunchKernelGGL(ep_rowln_apply_kernel, dim3(eg), dim3(256), 0, st, w.c2, w.statf, pr.norm_w, pr.norm_b, nd, D, out);
  EP_LAUNCH_CHECK("ep_cait forward kernels");
  return 0;
}

static int cait_backward_core(const ep_cait_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                              const float* tokstat, const ep_cait_params& pr, const float* dout, const ep_cait_params& gr, int acc,
                              const CaitWs& w, SideTasks sd, hipStream_t st, hipStream_t aux) {
  const int D = d.D, dh = D / d.H, Hd = d.hidden, B = d.B, H = d.H;
  const float scale = (float)pow((double)dh, -0.5);
  const int64_t n4 = (int64_t)B * Hd / 4;
  const unsigned cgrid = (D + CG - 1) / CG;
  if (!tokstat) tokstat = w.tstat;
  // the weight-gradient contractions: on the aux stream, each as early as its operands exist (AuxSide, ep_internal.h)
  PoolParams p = cait_pool_params(d, x, x_dtype, bstride, index, tokstat, w);
  p.ML = w.ML2;                                       // the (N + 1)-entry softmax state
  p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
  const bool in_pass = pool_backward_takes_side(p);   // side workgroups of the second pass where its kernel takes them
  AuxSide ax;
  EP_TRY(aux_side_begin(ax, st, in_pass ? nullptr : aux));
  EP_TRY(aux_side_fork(ax, sd));                     // the caller's (the classifier's weight gradient)
  GemmParams gW2 = cgm(w.dm2, D, w.h1, Hd, gr.fc2_w, Hd, D, Hd, B); gW2.accumulate = acc; gW2.side = 1;
  GemmParams gW1 = cgm(w.dh1, Hd, w.h2, D, gr.fc1_w, D, Hd, D, B); gW1.accumulate = acc; gW1.side = 1;
  GemmParams gWp = cgm(w.dz1, D, w.ya, D, gr.proj_w, D, D, D, B); gWp.accumulate = acc; gWp.side = 1;
  GemmParams gWv = cgm(w.dya0, D, w.P, (int64_t)H * D, w.dWvs, D, dh, D, B);                       // (rho dO)_h^T Phat_h
  gWv.sAz = dh; gWv.extA = dh; gWv.sBz = D; gWv.extB = D; gWv.sCz = (int64_t)dh * D; gWv.side = 1;
  EP_REQUIRE(gemm_side_ok(gW2, false, false) && gemm_side_ok(gW1, false, false) && gemm_side_ok(gWp, false, false) &&
             gemm_side_ok(gWv, false, false), EP_E_ALIGN, "cait: unaligned gradient contraction");
  // out = norm(c2)
  hipLaunchKernelGGL(ep_rowln_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, dout, w.c2, w.statf, pr.norm_w, (const float*)nullptr,
                     B, D, w.dc2);
  EP_TRY(lnaffine_grad(dout, w.c2, w.statf, B, D, acc, gr.norm_w,
                     gr.norm_b, st));
  // c2 = c1 + gamma_2 * m2 ;  m2 = fc2(gelu(fc1(norm2(c1))))
  hipLaunchKernelGGL(ep_cait_scale_bwd_kernel, dim3(cgrid), dim3(256), 0, st, w.dc2, w.m2, pr.gamma_2, B, D, acc, w.dm2, gr.gamma_2,
                     (float*)nullptr);
  EP_LAUNCH_CHECK("ep_cait final-norm backward kernels");
  if (!side_add_colsum(sd, w.dm2, B, D, D, acc, gr.fc2_b)) EP_TRY(colsum(w.dm2, B, D, D, acc, gr.fc2_b, st));
  side_add_gemm(sd, gW2, 1);
  EP_TRY(aux_side_fork(ax, sd));                     // dW2 = dm2^T h1
  EP_TRY(gemm(true, false, cgm(w.dm2, D, pr.fc2_w, Hd, w.dh1, Hd, B, Hd, D), 1, st));              // dh1 = dm2 W2
  hipLaunchKernelGGL(ep_gelu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.pre, n4, w.dh1);   // -> dpre
  EP_LAUNCH_CHECK("ep_gelu_bwd_kernel");
  if (!side_add_colsum(sd, w.dh1, B, Hd, Hd, acc, gr.fc1_b)) EP_TRY(colsum(w.dh1, B, Hd, Hd, acc, gr.fc1_b, st));
  side_add_gemm(sd, gW1, 1);
  EP_TRY(aux_side_fork(ax, sd));                     // dW1 = dpre^T h2
  { GemmParams g = cgm(w.dh1, Hd, pr.fc1_w, D, w.dh2, D, B, D, Hd); g.skws = w.skws; g.skws_floats = w.skws_floats;
    EP_TRY(gemm(true, false, g, 1, st)); }                                                          // dh2 = dpre W1
  hipLaunchKernelGGL(ep_rowln_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, w.dh2, w.c1, w.stat2, pr.n2_w, w.dc2, B, D, w.dc1);
  EP_TRY(lnaffine_grad(w.dh2, w.c1, w.stat2, B, D, acc, gr.n2_w,
                     gr.n2_b, st));
  // c1 = c + gamma_1 * z1 ;  z1 = ya Wp^T + bp          (dcsum = sum_b dc1: the direct path to the class token)
  hipLaunchKernelGGL(ep_cait_scale_bwd_kernel, dim3(cgrid), dim3(256), 0, st, w.dc1, w.z1, pr.gamma_1, B, D, acc, w.dz1, gr.gamma_1,
                     w.dcsum);
  EP_LAUNCH_CHECK("ep_cait block backward kernels");
  if (!side_add_colsum(sd, w.dz1, B, D, D, acc, gr.proj_b)) EP_TRY(colsum(w.dz1, B, D, D, acc, gr.proj_b, st));
  side_add_gemm(sd, gWp, 1);
  EP_TRY(aux_side_fork(ax, sd));                     // dWp = dz1^T ya
  EP_TRY(gemm(true, false, cgm(w.dz1, D, pr.proj_w, D, w.dya, D, B, D, D), 1, st));                // dya = dz1 Wp
  EP_TRY(colsum(w.dya, B, D, D, 0, w.dbo, st));                                                    // d(Wv b1 + bv)
  EP_TRY(delta_rows(w.dya, w.ya, B * H, dh, w.ML2, st, w.bo, H));                                  // delta' = dO . (o - bias)
  hipLaunchKernelGGL(ep_cait_clsgrad_kernel, dim3((B * H + 3) / 4), dim3(256), 0, st, w.dya, w.vc, w.ML2, w.mix, B * H, dh, H,
                     w.dya0, w.csc);
  side_add_gemm(sd, gWv, H);
  EP_TRY(aux_side_fork(ax, sd));                     // dWv' = dya0^T Phat
  EP_TRY(aux_side_rest(ax, sd));                     // the column sums, the statistics
  hipLaunchKernelGGL(ep_cait_dvc_kernel, dim3(cgrid), dim3(256), 0, st, w.dya, w.mix, B, D, dh, H, w.dvc);
  EP_LAUNCH_CHECK("ep_cait class-entry kernels");
  {
    GemmParams g = cgm(w.dya, D, w.Wvs, D, w.dP, (int64_t)H * D, B, D, dh);                         // dPhat'[b,h] = dO Wv'_h
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D;
    EP_TRY(gemm(true, false, g, H, st));
  }
  if (in_pass) {
    EP_TRY(pool_backward(p, w.dw, 0, st, &sd));
  } else {
    EP_TRY(aux_side_before_pass(ax, sd));
    EP_TRY(pool_backward(p, w.dw, 0, st));
    EP_TRY(aux_side_join(ax));
  }
  // class entry: dw += chat sum_b dS_c ; dchat ; dWv' += dvc chat^T
  hipLaunchKernelGGL(ep_cait_clsred_part_kernel, dim3((D + 31) / 32, CAIT_RS), dim3(1024), 0, st, w.csc, w.dP, (int64_t)B * H, D, w.cpart);
  hipLaunchKernelGGL(ep_cait_clsred_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.csc, w.cpart, CAIT_RS, w.wq, w.chat, B, H, D, w.dw,
                     w.dchat);
  hipLaunchKernelGGL(ep_cait_rank1_kernel, dim3((D + 255) / 256, D), dim3(256), 0, st, w.dvc, w.chat, D, w.dWvs);
  // value side: d v.weight, and the value-side parts of d norm1.weight / bias ; d v.bias = dbo
  hipLaunchKernelGGL(ep_cae_dwv_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.dWvs, w.dbo, pr.v_w, pr.n1_w, pr.n1_b, D, acc,
                     gr.v_w, gr.n1_w, gr.n1_b, (float*)nullptr, (float*)nullptr);
  EP_LAUNCH_CHECK("ep_cait value backward kernels");
  EP_TRY(colsum(w.dya, B, D, D, acc, gr.v_b, st));
  // key side: du = g1 * dw ; d norm1.weight += sum_h u_h * dw_h
  hipLaunchKernelGGL(ep_cae_du_kernel, dim3((D + 255) / 256), dim3(256), 0, st, w.dw, w.u, pr.n1_w, D, H, 1, w.du, gr.n1_w, gr.n1_b);
  // query chain: d q.weight / bias, d k.weight, d k.bias = 0, dun0 = Wq^T dq
  hipLaunchKernelGGL(ep_siglip_dq_kernel, dim3((D + 3) / 4), dim3(256), 0, st, w.du, pr.k_w, D, dh, scale, acc, w.dq, gr.q_b);
  if (acc) EP_HIP(hipMemsetAsync(w.dun0, 0, (size_t)D * sizeof(float), st));   // (the kernel accumulates all its outputs alike)
  EP_TRY(siglip_qgrad(w.q, w.dq, w.du,
                     w.un0, pr.q_w, D, dh, scale, acc, gr.k_w, gr.q_w, w.dun0, gr.k_b, st));
  // class token: LayerNorm backward with both gradient kinds + the direct residual path
  hipLaunchKernelGGL(ep_cait_clsln_bwd_kernel, dim3(1), dim3(256), 0, st, w.dun0, w.dchat, w.chat, pr.n1_w, w.lnstat, w.dcsum, D, acc,
                     gr.n1_w, gr.n1_b, gr.cls_token);
  EP_LAUNCH_CHECK("ep_cait backward kernels");
  return 0;
}

static ep_cait_params cait_views(float* base, const int64_t o[CAIT_NT]) {
  ep_cait_params p;
  p.cls_token = base + o[0]; p.gamma_1 = base + o[1]; p.gamma_2 = base + o[2]; p.n1_w = base + o[3]; p.n1_b = base + o[4];
  p.q_w = base + o[5]; p.q_b = base + o[6]; p.k_w = base + o[7]; p.k_b = base + o[8]; p.v_w = base + o[9]; p.v_b = base + o[10];
  p.proj_w = base + o[11]; p.proj_b = base + o[12]; p.n2_w = base + o[13]; p.n2_b = base + o[14]; p.fc1_w = base + o[15];
  p.fc1_b = base + o[16]; p.fc2_w = base + o[17]; p.fc2_b = base + o[18]; p.norm_w = base + o[19]; p.norm_b = base + o[20];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

size_t ep_cait_pool_workspace_bytes(const ep_cait_dims* dims) {
  if (!dims || cait_check(*dims, false) != 0) return 0;
  return cait_carve(*dims, nullptr, false).total;
}

int ep_cait_pool_forward(const ep_cait_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                         const float* token_stats_, const ep_cait_params* params, float* out, void* ws, size_t ws_bytes,
                         ep_stream_t stream) {
  EP_REQUIRE(dims && out && ws, EP_E_ARG, "ep_cait_pool_forward: null pointer");
  EP_TRY(cait_check(*dims, false));
  EP_TRY(cait_params_ok(params, "ep_cait_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(out), EP_E_ALIGN, "ep_cait_pool_forward: out / ws must be 16-byte aligned");
  const CaitWs w = cait_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cait_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return cait_forward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, *params, w, out, (hipStream_t)stream);
}

int ep_cait_pool_backward(const ep_cait_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                          const float* token_stats_, const ep_cait_params* params, const float* dout,
                          const ep_cait_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dout && ws, EP_E_ARG, "ep_cait_pool_backward: null pointer");
  EP_TRY(cait_check(*dims, false));
  EP_TRY(cait_params_ok(params, "ep_cait_pool_backward(params)"));
  EP_TRY(cait_params_ok(grads, "ep_cait_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dout), EP_E_ALIGN, "ep_cait_pool_backward: dout / ws must be 16-byte aligned");
  const CaitWs w = cait_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cait_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return cait_backward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, *params, dout, *grads, accumulate, w,
                            SideTasks{}, (hipStream_t)stream, nullptr);
}

int64_t ep_cait_head_param_offsets(const ep_cait_dims* dims, int64_t offsets[23]) { return cait_offsets(*dims, offsets); }

size_t ep_cait_head_workspace_bytes(const ep_cait_dims* dims) {
  if (!dims || cait_check(*dims, true) != 0) return 0;
  return cait_carve(*dims, nullptr, true).total;
}

int ep_cait_head_train_step(const ep_cait_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_cait_head_train_step: null pointer");
  const ep_cait_dims& d = s->dims;
  EP_TRY(cait_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const CaitWs w = cait_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cait_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CAIT_NT];
  const int64_t total = cait_offsets(d, offs);
  const ep_cait_params pr = cait_views(s->params, offs), gr = cait_views(s->grads, offs);
  float* Wc = s->params + offs[21]; float* bc = s->params + offs[22];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    EP_TRY(cait_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, nullptr, nullptr, 0, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    SideTasks sd{};
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, d.D, d.C, s->grads + offs[21], s->accumulate);
    EP_REQUIRE(gemm_side_ok(gWc, false, false), EP_E_ALIGN, "cait head: unaligned classifier gradient");
    side_add_gemm(sd, gWc, 1);
    sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[22]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
    sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
    sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
    sd.total += sd.n_colsum + sd.n_stats;
    EP_TRY(cait_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, pr, w.dy, gr, s->accumulate, w, sd,
                              st, (hipStream_t)s->aux_stream));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    int64_t sizes[CAIT_NT];
    cait_sizes(d, sizes);
    // util/lars.py:22: trust ratio + weight decay for ndim > 1: cls_token is (1,1,D); the LayerScale vectors, LayerNorm
    // parameters and biases are one-dimensional
    const int trust[CAIT_NT] = {1, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0};
    ep_segment segs[CAIT_NT];
    for (int i = 0; i < CAIT_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? CAIT_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_cait_head_eval_forward(const ep_cait_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                              const float* token_stats_, const float* params, const float* running_mean,
                              const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                              ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_cait_head_eval_forward: null pointer");
  const ep_cait_dims& d = *dims;
  EP_TRY(cait_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const CaitWs w = cait_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cait_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CAIT_NT];
  cait_offsets(d, offs);
  const ep_cait_params pr = cait_views(const_cast<float*>(params), offs);
  EP_TRY(cait_forward_core(d, x, x_dtype, x_bstride, image_index, token_stats_, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[21], params + offs[22], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
