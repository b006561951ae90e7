// CoCa attentional pooler (reference poolings/coca_pytorch.py:250-343, registry entry probe_heads.py:78)
// on the EP streaming kernels.
//
// The reference layer-norms all M image queries, projects them to H heads, attends over one shared
// key/value head and returns only query 0.  With q0 = scale * to_q(LN(img_queries[0])) (H x dh):
//     sim[b,h,n] = q0[h] . (Wk x[b,n])            = u[h] . x[b,n],   u[h] = Wk^T q0[h]   (D floats)
//     out[b,h]   = sum_n softmax_n(sim)[n] (Wv x[b,n]) = (sum_n A[b,h,n] x[b,n]) Wv^T = P[b,h] Wv^T
// so the token-dependent part is exactly the EP pooling pass with H query rows u (scale already folded
// in), followed by two small contractions (shared Wv, then to_out).  The backward needs du only; the
// chain from du back to to_kv[:dh], to_q, the LayerNorm gain and img_queries[0] is batch independent
// (a few hundred thousand FMAs) and runs in three tiny kernels.
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_lnaffine.h"

namespace ep {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// qn = LayerNorm(img_queries[0]) (coca_pytorch.py:70-77: F.layer_norm, biased variance, eps inside the
// root); qh[j] = scale * to_q.weight[j] . qn (coca_pytorch.py:310-316).  One wave per output row; every
// workgroup recomputes the D-float LayerNorm, workgroup 0 saves it for the backward.
__global__ __launch_bounds__(256) void ep_coca_q_kernel(const float* __restrict__ imgq, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ Wq,
                                                      int D, int HD, float eps, float scale, float* __restrict__ xhat,
                                                      float* __restrict__ qn, float* __restrict__ lnstat,
                                                      float* __restrict__ qh) {
  extern __shared__ float sq[];          // D floats: qn
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int d = tid; d < D; d += 256) s += imgq[d];
  const float mean = block_sum_256(s, red) / (float)D;
  float v = 0.f;
  for (int d = tid; d < D; d += 256) { const float c = imgq[d] - mean; v = fmaf(c, c, v); }
  const float rstd = 1.0f / sqrtf(block_sum_256(v, red) / (float)D + eps);
  for (int d = tid; d < D; d += 256) {
    const float xh = (imgq[d] - mean) * rstd;
    const float q = fmaf(xh, gamma[d], beta ? beta[d] : 0.f);
    sq[d] = q;
    if (blockIdx.x == 0) { xhat[d] = xh; qn[d] = q; }
  }
  if (blockIdx.x == 0 && tid == 0) { lnstat[0] = mean; lnstat[1] = rstd; }
  __syncthreads();
  const int j = blockIdx.x * 4 + (tid >> 6);
  if (j >= HD) return;
  const int lane = tid & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(Wq[(int64_t)j * D + d], sq[d], acc);
  acc = wave_sum(acc);
  if (lane == 0) qh[j] = acc * scale;
}

// u[h,d] = sum_c qh[h*dh + c] * Wk[c,d]   (Wk = to_kv.weight[0:dh], coca_pytorch.py:320)
__global__ __launch_bounds__(1024) void ep_coca_u_kernel(const float* __restrict__ qh, const float* __restrict__ Wk,
                                                       int D, int dh, float* __restrict__ u) {
  __shared__ float sm[32][33];                       // grid (D / 32, H): 32 column lanes x 32 row lanes over the dh key rows
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx, h = blockIdx.y;
  float acc = 0.f;
  if (d < D)
    for (int c = ty; c < dh; c += 32) acc = fmaf(qh[h * dh + c], Wk[(int64_t)c * D + d], acc);
  sm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    u[(int64_t)h * D + d] = t;
  }
}

// dqh[h*dh + c] = du[h] . Wk[c]
__global__ __launch_bounds__(256) void ep_coca_dqh_kernel(const float* __restrict__ du, const float* __restrict__ Wk,
                                                        int D, int dh, int HD, float* __restrict__ dqh) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= HD) return;
  const int h = j / dh, c = j % dh, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(du[(int64_t)h * D + d], Wk[(int64_t)c * D + d], acc);
  acc = wave_sum(acc);
  if (lane == 0) dqh[j] = acc;
}

// Per 64-column block of d:  dWk[c,d] (+)= sum_h qh[h,c] du[h,d];  dWq[j,d] (+)= scale dqh[j] qn[d];
// dqn[d] = scale sum_j Wq[j,d] dqh[j];  rows 1..M-1 of d img_queries <- 0 (no gradient path) unless accumulating.
// (two launches: the row-wise parts on grid (D / 64, 8), every row range split eight ways, and the (1 x HD) . (HD x D)
// product for d qn on grid D / 32 with 1024 threads = 32 column lanes x 32 row lanes)
__global__ __launch_bounds__(256) void ep_coca_qgrad_kernel(const float* __restrict__ qh, const float* __restrict__ dqh,
                                                          const float* __restrict__ du, const float* __restrict__ qn, int D, int H,
                                                          int dh, int M, float scale, int accumulate, float* __restrict__ dWk,
                                                          float* __restrict__ dWq, float* __restrict__ dimgq) {
  const int HD = H * dh;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int d = blockIdx.x * 64 + tx;
  if (d >= D) return;
  const int gy = gridDim.y, by = blockIdx.y;
  auto range = [&](int n, int& r0, int& r1) { const int per = (n + gy - 1) / gy; r0 = by * per; r1 = (r0 + per) < n ? (r0 + per) : n; };
  int r0, r1;
  range(dh, r0, r1);
  for (int c = r0 + ty; c < r1; c += 4) {
    float g = 0.f;
    for (int h = 0; h < H; ++h) g = fmaf(qh[h * dh + c], du[(int64_t)h * D + d], g);
    float* o = dWk + (int64_t)c * D + d;
    *o = accumulate ? *o + g : g;
  }
  const float qd = qn[d] * scale;
  range(HD, r0, r1);
  for (int j = r0 + ty; j < r1; j += 4) {
    float* o = dWq + (int64_t)j * D + d;
    const float g = dqh[j] * qd;
    *o = accumulate ? *o + g : g;
  }
  if (!accumulate) {
    range(M - 1, r0, r1);
    for (int r = 1 + r0 + ty; r < 1 + r1; r += 4) dimgq[(int64_t)r * D + d] = 0.f;
  }
}
__global__ __launch_bounds__(1024) void ep_coca_dqn_kernel(const float* __restrict__ dqh, const float* __restrict__ Wq, int D, int HD,
                                                         float scale, float* __restrict__ dqn) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (d < D)
    for (int j = ty; j < HD; j += 32) acc = fmaf(Wq[(int64_t)j * D + d], dqh[j], acc);
  sm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    dqn[d] = t * scale;
  }
}

// LayerNorm backward of the single live row: dgamma (+)= dqn * xhat; d img_queries[0] (+)= the usual
// rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dqn * gamma.
__global__ __launch_bounds__(256) void ep_coca_lnbwd_kernel(const float* __restrict__ dqn, const float* __restrict__ xhat,
                                                          const float* __restrict__ gamma, const float* __restrict__ lnstat,
                                                          int D, int accumulate, float* __restrict__ dgamma,
                                                          float* __restrict__ dimgq0, float* __restrict__ dbeta) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (int d = tid; d < D; d += 256) { const float g = dqn[d] * gamma[d]; a += g; b = fmaf(g, xhat[d], b); }
  const float m1 = block_sum_256(a, red) / (float)D;
  const float m2 = block_sum_256(b, red) / (float)D;
  const float rstd = lnstat[1];
  for (int d = tid; d < D; d += 256) {
    const float g = dqn[d] * gamma[d];
    const float dx = rstd * (g - m1 - xhat[d] * m2);
    const float dg = dqn[d] * xhat[d];
    dgamma[d] = accumulate ? dgamma[d] + dg : dg;
    dimgq0[d] = accumulate ? dimgq0[d] + dx : dx;
    if (dbeta) dbeta[d] = accumulate ? dbeta[d] + dqn[d] : dqn[d];
  }
}

// ---------------------------------------------------------------------------------------------
struct CocaWs {
  float *P, *S, *ML, *o, *dO, *dP, *xhat, *qn, *qh, *u, *lnstat, *du, *dqh, *dqn, *dWvp;
  void* pool_ws; size_t pool_ws_bytes;
  size_t pool_total;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static int64_t coca_offsets(const ep_coca_dims& d, int64_t offs[7]) {
  const int64_t HD = (int64_t)d.H * d.dh;
  const int64_t sizes[7] = {d.D, (int64_t)d.M * d.D, HD * d.D, 2LL * d.dh * d.D, (int64_t)d.D * HD,
                            (int64_t)d.C * d.D, d.C};
  int64_t off = 0;
  for (int i = 0; i < 7; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static CocaWs coca_carve(const ep_coca_dims& d, void* base, bool head) {
  CocaWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, HD = (size_t)d.H * d.dh;
  w.P = take(B * d.H * d.D); w.S = take(B * d.H * d.N); w.ML = take(B * d.H * 4);
  w.o = take(B * HD); w.dO = take(B * HD); w.dP = take(B * d.H * d.D);
  w.xhat = take(d.D); w.qn = take(d.D); w.qh = take(HD); w.u = take((size_t)d.H * d.D); w.lnstat = take(4);
  w.du = take((size_t)d.H * d.D); w.dqh = take(HD); w.dqn = take(d.D); w.dWvp = take(HD * d.D);
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.H);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  w.pool_total = off;
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * d.D); w.z = take(B * d.D); w.rstd = take(d.D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * d.D); w.dy = take(B * d.D);
    int64_t offs[7];
    w.opt_ws_bytes = optim_workspace_bytes(coca_offsets(d, offs), 7);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int coca_check(const ep_coca_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0 && d.dh > 0 && d.M > 0, EP_E_ARG, "coca dims must be positive");
  EP_REQUIRE(d.D % 4 == 0 && d.dh % 4 == 0, EP_E_SHAPE, "coca: D and dim_head must be multiples of 4 (D=%d dh=%d)", d.D, d.dh);
  EP_REQUIRE(d.H <= 32, EP_E_UNSUPPORTED, "coca: heads = %d > 32", d.H);
  EP_REQUIRE((size_t)(2 * d.H * d.dh + 256) * 4 <= 60000 && (size_t)d.D * 4 <= 60000, EP_E_UNSUPPORTED, "coca: D / inner dim too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "coca head: C must be positive");
  return 0;
}

static int coca_params_ok(const ep_coca_params* p, const char* what) {
  EP_REQUIRE(p && p->gamma && p->img_queries && p->to_q && p->to_kv && p->to_out, EP_E_ARG, "%s: null tensor", what);
  EP_REQUIRE(aligned16(p->gamma) && aligned16(p->img_queries) && aligned16(p->to_q) && aligned16(p->to_kv) &&
             aligned16(p->to_out), EP_E_ALIGN, "%s: tensors must be 16-byte aligned", what);
  return 0;
}

static PoolParams coca_pool_params(const ep_coca_dims& d, const void* x, int64_t bstride, const int32_t* index,
                                   const CocaWs& w, int x_dtype) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);     // the scale lives in u
  p.cls = w.u; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index;
  return p;
}

// y (B,D) = to_out(concat_h(P[b,h] Wv^T))
static int coca_forward_core(const ep_coca_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const ep_coca_params& pr, float ln_eps, const CocaWs& w, float* y, hipStream_t st) {
  const int D = d.D, HD = d.H * d.dh;
  const float scale = (float)pow((double)d.dh, -0.5);                   // coca_pytorch.py:266
  hipLaunchKernelGGL(ep_coca_q_kernel, dim3((HD + 3) / 4), dim3(256), (size_t)D * 4, st, pr.img_queries, pr.gamma,
                     pr.beta, pr.to_q, D, HD, ln_eps, scale, w.xhat, w.qn, w.lnstat, w.qh);
  hipLaunchKernelGGL(ep_coca_u_kernel, dim3((D + 31) / 32, d.H), dim3(1024), 0, st, w.qh, pr.to_kv, D, d.dh, w.u);
  EP_LAUNCH_CHECK("ep_coca_q/u kernels");
  EP_TRY(pool_forward(coca_pool_params(d, x, bstride, index, w, x_dtype), st));
  const float* Wv = pr.to_kv + (int64_t)d.dh * D;
  GemmParams g{};                                                       // o[b, h*dh + c] = P[b,h,:] . Wv[c,:]
  g.A = w.P; g.lda = (int64_t)d.H * D; g.sAz = D; g.extA = D;
  g.B = Wv; g.ldb = D; g.sBz = 0; g.extB = D;
  g.C = w.o; g.ldc = HD; g.sCz = d.dh;
  g.M = d.B; g.N = d.dh; g.K = D; g.alpha = 1.f;
  EP_TRY(gemm(true, true, g, d.H, st));
  GemmParams h{};                                                       // y = o to_out^T
  h.A = w.o; h.lda = HD; h.B = pr.to_out; h.ldb = HD; h.C = y; h.ldc = D;
  h.M = d.B; h.N = D; h.K = HD; h.alpha = 1.f; h.extA = HD; h.extB = HD;
  return gemm(true, true, h, 1, st);
}

// gradients of the five pooler tensors from dy (B,D).  `extra`: more side work for the token pass
// (the classifier's weight gradients when called from the whole-head step).
static int coca_backward_core(const ep_coca_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                              const ep_coca_params& pr, const float* dy, const ep_coca_params& gr, int accumulate,
                              const CocaWs& w, SideTasks sd, hipStream_t st, hipStream_t aux) {
  const int D = d.D, HD = d.H * d.dh;
  const float scale = (float)pow((double)d.dh, -0.5);
  const float* Wv = pr.to_kv + (int64_t)d.dh * D;
  {
    GemmParams g{};                                                     // dO = dy to_out
    g.A = dy; g.lda = D; g.B = pr.to_out; g.ldb = HD; g.extB = HD; g.C = w.dO; g.ldc = HD;
    g.M = d.B; g.N = HD; g.K = D; g.alpha = 1.f;
    EP_TRY(gemm(true, false, g, 1, st));
  }
  EP_TRY(delta_rows(w.dO, w.o, d.B * d.H, d.dh, w.ML, st));              // softmax correction term, = dP . P
  {
    GemmParams g{};                                                     // dP[b,h,:] = dO[b,h,:] Wv
    g.A = w.dO; g.lda = HD; g.sAz = d.dh; g.B = Wv; g.ldb = D; g.sBz = 0; g.extB = D;
    g.C = w.dP; g.ldc = (int64_t)d.H * D; g.sCz = D;
    g.M = d.B; g.N = D; g.K = d.dh; g.alpha = 1.f;
    EP_TRY(gemm(true, false, g, d.H, st));
  }
  GemmParams gWo{};                                                     // d to_out (D, HD) (+)= dy^T o
  gWo.A = dy; gWo.lda = D; gWo.extA = D; gWo.B = w.o; gWo.ldb = HD; gWo.extB = HD; gWo.C = gr.to_out; gWo.ldc = HD;
  gWo.M = D; gWo.N = HD; gWo.K = d.B; gWo.alpha = 1.f; gWo.accumulate = accumulate; gWo.side = 1;
  GemmParams gWv{};                                                     // per-head partials of dWv (dh, D)
  gWv.A = w.dO; gWv.lda = HD; gWv.sAz = d.dh; gWv.extA = d.dh;
  gWv.B = w.P; gWv.ldb = (int64_t)d.H * D; gWv.sBz = D; gWv.extB = D;
  gWv.C = w.dWvp; gWv.ldc = D; gWv.sCz = (int64_t)d.dh * D;
  gWv.M = d.dh; gWv.N = D; gWv.K = d.B; gWv.alpha = 1.f; gWv.side = 1;
  EP_REQUIRE(gemm_side_ok(gWo, false, false) && gemm_side_ok(gWv, false, false), EP_E_ALIGN, "coca: unaligned gradient contraction");
  side_add_gemm(sd, gWo, 1);
  side_add_gemm(sd, gWv, d.H);
  PoolParams p = coca_pool_params(d, x, bstride, index, w, x_dtype);
  p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
  if (pool_backward_takes_side(p)) {
    EP_TRY(pool_backward(p, w.du, 0, st, &sd));
  } else {                                           // (kernel families without side workgroups: the aux stream)
    AuxSide ax;
    EP_TRY(aux_side_begin(ax, st, aux));
    EP_TRY(aux_side_before_pass(ax, sd));
    EP_TRY(pool_backward(p, w.du, 0, st));
    EP_TRY(aux_side_join(ax));
  }
  // dWv = sum over heads of the partials -> rows dh..2dh-1 of d to_kv
  EP_TRY(reduce_partials(w.dWvp, d.H, d.dh * D, 1.0f, accumulate, gr.to_kv + (int64_t)d.dh * D, nullptr, st));
  hipLaunchKernelGGL(ep_coca_dqh_kernel, dim3((HD + 3) / 4), dim3(256), 0, st, w.du, pr.to_kv, D, d.dh, HD, w.dqh);
  hipLaunchKernelGGL(ep_coca_qgrad_kernel, dim3((D + 63) / 64, 8), dim3(256), 0, st, w.qh, w.dqh, w.du, w.qn, D, d.H, d.dh, d.M,
                     scale, accumulate, gr.to_kv, gr.to_q, gr.img_queries);
  hipLaunchKernelGGL(ep_coca_dqn_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.dqh, pr.to_q, D, HD, scale, w.dqn);
  hipLaunchKernelGGL(ep_coca_lnbwd_kernel, dim3(1), dim3(256), 0, st, w.dqn, w.xhat, pr.gamma, w.lnstat, D, accumulate,
                     gr.gamma, gr.img_queries, (float*)nullptr);
  EP_LAUNCH_CHECK("ep_coca backward kernels");
  return 0;
}

}  // namespace ep

using namespace ep;

extern "C" {

size_t ep_coca_pool_workspace_bytes(const ep_coca_dims* dims) {
  if (!dims || coca_check(*dims, false) != 0) return 0;
  return coca_carve(*dims, nullptr, false).total;
}

int ep_coca_pool_forward(const ep_coca_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const ep_coca_params* params, float ln_eps, float* y, void* ws,
                         size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_coca_pool_forward: null pointer");
  EP_TRY(coca_check(*dims, false));
  EP_TRY(coca_params_ok(params, "ep_coca_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_coca_pool_forward: y / ws must be 16-byte aligned");
  const CocaWs w = coca_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_coca_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return coca_forward_core(*dims, x, x_dtype, x_bstride, image_index, *params, ln_eps, w, y, (hipStream_t)stream);
}

int ep_coca_pool_backward(const ep_coca_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const int32_t* image_index, const ep_coca_params* params, const float* dy,
                          const ep_coca_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dy && ws, EP_E_ARG, "ep_coca_pool_backward: null pointer");
  EP_TRY(coca_check(*dims, false));
  EP_TRY(coca_params_ok(params, "ep_coca_pool_backward(params)"));
  EP_TRY(coca_params_ok(grads, "ep_coca_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dy), EP_E_ALIGN, "ep_coca_pool_backward: dy / ws must be 16-byte aligned");
  const CocaWs w = coca_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_coca_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return coca_backward_core(*dims, x, x_dtype, x_bstride, image_index, *params, dy, *grads, accumulate, w, SideTasks{},
                            (hipStream_t)stream, nullptr);
}

int ep_coca_attention(const ep_coca_dims* dims, const void* ws, float* A, ep_stream_t stream) {
  EP_REQUIRE(dims && ws && A, EP_E_ARG, "ep_coca_attention: null pointer");
  EP_TRY(coca_check(*dims, false));
  const CocaWs w = coca_carve(*dims, const_cast<void*>(ws), false);
  return attention_from_scores(w.S, w.ML, dims->B * dims->H, dims->N, A, (hipStream_t)stream);
}

int64_t ep_coca_head_param_offsets(const ep_coca_dims* dims, int64_t offsets[7]) { return coca_offsets(*dims, offsets); }

size_t ep_coca_head_workspace_bytes(const ep_coca_dims* dims) {
  if (!dims || coca_check(*dims, true) != 0) return 0;
  return coca_carve(*dims, nullptr, true).total;
}

int64_t ep_coca_head_workspace_logits_offset(const ep_coca_dims* dims, int32_t* ldl) {
  if (!dims || coca_check(*dims, true) != 0) return -1;
  char* base = reinterpret_cast<char*>(uintptr_t(1) << 20);   // coca_carve() only does address arithmetic on a non-null base
  const CocaWs w = coca_carve(*dims, base, true);
  if (ldl) *ldl = w.ldl;
  return reinterpret_cast<char*>(w.logits) - base;
}

static ep_coca_params coca_views(float* base, const int64_t offs[7], const float* beta) {
  ep_coca_params p;
  p.gamma = base + offs[0]; p.beta = beta; p.img_queries = base + offs[1]; p.to_q = base + offs[2];
  p.to_kv = base + offs[3]; p.to_out = base + offs[4];
  return p;
}

int ep_coca_head_train_step(const ep_coca_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_coca_head_train_step: null pointer");
  const ep_coca_dims& d = s->dims;
  EP_TRY(coca_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const CocaWs w = coca_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_coca_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  EP_REQUIRE(s->arith == EP_ARITH_F32 || s->arith == EP_ARITH_BF16_AUTOCAST, EP_E_ARG, "ep_coca_head_train_step: arith %d", s->arith);
  const ArithScope arith_scope(s->arith);            // AMP-bf16: the contractions below as one bf16 product (ep_gemm.hip: gemm_b3_ok)
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[7];
  const int64_t total = coca_offsets(d, offs);
  const ep_coca_params pr = coca_views(s->params, offs, s->ln_beta);
  const ep_coca_params gr = coca_views(s->grads, offs, nullptr);
  float* Wc = s->params + offs[5]; float* bc = s->params + offs[6];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    EP_TRY(coca_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, pr, s->ln_eps, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, nullptr, nullptr, 0, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    // classifier weight / bias gradients and the statistics fold ride in the second token pass
    SideTasks sd{};
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, d.D, d.C, s->grads + offs[5], s->accumulate);
    EP_REQUIRE(gemm_side_ok(gWc, false, false), EP_E_ALIGN, "coca head: unaligned classifier gradient");
    side_add_gemm(sd, gWc, 1);
    sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[6]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
    sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
    sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
    sd.total += sd.n_colsum + sd.n_stats;
    EP_TRY(coca_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, pr, w.dy, gr, s->accumulate, w, sd, st,
                              (hipStream_t)s->aux_stream));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t HD = (int64_t)d.H * d.dh;
    const int64_t sizes[7] = {d.D, (int64_t)d.M * d.D, HD * d.D, 2LL * d.dh * d.D, (int64_t)d.D * HD,
                              (int64_t)d.C * d.D, d.C};
    ep_segment segs[7];
    for (int i = 0; i < 7; ++i) segs[i] = ep_segment{offs[i], sizes[i], (i == 0 || i == 6) ? 0 : 1, 0};   // 1-D tensors: no trust ratio
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? 7 : 0, s->lr, s->weight_decay,
                      s->momentum, s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step,
                      s->found_inf, s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_coca_head_eval_forward(const ep_coca_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                              const int32_t* image_index, const float* params, const float* ln_beta, float ln_eps,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits,
                              int ldl, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_coca_head_eval_forward: null pointer");
  const ep_coca_dims& d = *dims;
  EP_TRY(coca_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const CocaWs w = coca_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_coca_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[7];
  coca_offsets(d, offs);
  const ep_coca_params pr = coca_views(const_cast<float*>(params), offs, ln_beta);
  EP_TRY(coca_forward_core(d, x, x_dtype, x_bstride, image_index, pr, ln_eps, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[5], params + offs[6], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"

// =============================================================================================
// CAE attentive block (reference poolings/cae_att.py:79-108 CAEAttentiveBlock with its CrossAttention :19-77; registry
// entry probe_heads.py:83: CAEAttentiveBlock(dim=dim) -> 8 heads, no qkv bias).  One learned query token; the keys
// are LN_k(x) Wk^T and the values LN_v(x) Wv^T -- two LayerNorms of the SAME token, i.e. the same normalised token
// xhat with two affine maps.  With qh = scale Wq LN_q(query) and u_h = Wk_h^T qh_h:
//     score[b,h,n] = u_h . (gk * xhat[b,n] + bk) = (gk * u_h) . xhat[b,n] + const
//     o[b,h]       = Wv_h (gv * Phat[b,h] + bv),      Phat[b,h] = sum_n A[b,h,n] xhat[b,n]
// so the token-dependent part is the LayerNorm-of-tokens mode of the EP passes (PoolParams.tokstat) with the H query
// rows w_h = gk * u_h, followed by the per-head projection with Wv' = Wv diag(gv) and bias Wv bv, then proj.
// =============================================================================================
namespace ep {

// w[h,d] = gk[d] * u[h,d],  u[h,d] = sum_c qh[h*dh + c] Wk[h*dh + c, d]
__global__ __launch_bounds__(1024) void ep_cae_w_kernel(const float* __restrict__ qh, const float* __restrict__ Wk,
                                                      const float* __restrict__ gk, int D, int dh, float* __restrict__ u,
                                                      float* __restrict__ wq) {
  __shared__ float sm[32][33];                       // grid (D / 32, H): 32 column lanes x 32 row lanes over the head's dh rows
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx, h = blockIdx.y;
  float acc = 0.f;
  if (d < D)
    for (int c = ty; c < dh; c += 32) acc = fmaf(qh[h * dh + c], Wk[(int64_t)(h * dh + c) * D + d], acc);
  sm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    u[(int64_t)h * D + d] = t;
    wq[(int64_t)h * D + d] = t * gk[d];
  }
}

// dqh[j] = Wk[j,:] . du[h(j),:]
__global__ __launch_bounds__(256) void ep_cae_dqh_kernel(const float* __restrict__ du, const float* __restrict__ Wk, int D,
                                                       int dh, float* __restrict__ dqh) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D) return;
  const int h = j / dh, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(Wk[(int64_t)j * D + d], du[(int64_t)h * D + d], acc);
  acc = wave_sum(acc);
  if (lane == 0) dqh[j] = acc;
}

// per 64-column block of d:  dWk[j,d] (+)= qh[j] du[h(j),d];  dWq[j,d] (+)= scale dqh[j] qn[d];
// dqn[d] = scale sum_j Wq[j,d] dqh[j]
// (two launches: the outer products on grid (D / 64, 16) -- 64 columns x 4 row lanes over a sixteenth of the rows -- and
// the (1 x D) . (D x D) product for d qn on grid D / 32 with 1024 threads = 32 column lanes x 32 row lanes)
__global__ __launch_bounds__(256) void ep_cae_qgrad_kernel(const float* __restrict__ qh, const float* __restrict__ dqh,
                                                         const float* __restrict__ du, const float* __restrict__ qn, int D, int dh,
                                                         float scale, int accumulate, float* __restrict__ dWk,
                                                         float* __restrict__ dWq) {
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int d = blockIdx.x * 64 + tx;
  if (d >= D) return;
  const int per = (D + gridDim.y - 1) / gridDim.y;
  const int j0 = blockIdx.y * per, j1 = (j0 + per) < D ? (j0 + per) : D;
  const float qd = qn[d] * scale;
  for (int j = j0 + ty; j < j1; j += 4) {
    const float gk_ = qh[j] * du[(int64_t)(j / dh) * D + d];
    float* o1 = dWk + (int64_t)j * D + d;
    *o1 = accumulate ? *o1 + gk_ : gk_;
    const float gq = dqh[j] * qd;
    float* o2 = dWq + (int64_t)j * D + d;
    *o2 = accumulate ? *o2 + gq : gq;
  }
}
__global__ __launch_bounds__(1024) void ep_cae_dqn_kernel(const float* __restrict__ dqh, const float* __restrict__ Wq, int D,
                                                        float scale, float* __restrict__ dqn) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (d < D)
    for (int j = ty; j < D; j += 32) acc = fmaf(Wq[(int64_t)j * D + d], dqh[j], acc);
  sm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    dqn[d] = t * scale;
  }
}

constexpr int CAE_NT = 16;
struct CaeWs {
  float *P, *S, *ML, *o, *dO, *dP, *stats, *xhat, *qn, *qh, *u, *wq, *lnstat, *dw, *du, *dqh, *dqn, *Wvs, *bo, *dWvs, *dbo;
  void* pool_ws; size_t pool_ws_bytes;
  size_t pool_total;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static int64_t cae_offsets(const ep_cae_dims& d, int64_t offs[CAE_NT]) {
  const int64_t D = d.D;
  const int64_t sizes[CAE_NT] = {D, D, D, D, D, D, D, D, D, D * D, D * D, D * D, D * D, D, (int64_t)d.C * D, d.C};
  int64_t off = 0;
  for (int i = 0; i < CAE_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static CaeWs cae_carve(const ep_cae_dims& d, void* base, bool head) {
  CaeWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D;
  w.P = take(B * d.H * D); w.S = take(B * d.H * d.N); w.ML = take(B * d.H * 4);
  w.o = take(B * D); w.dO = take(B * D); w.dP = take(B * d.H * D); w.stats = take(B * d.N * 2);
  w.xhat = take(D); w.qn = take(D); w.qh = take(D); w.u = take((size_t)d.H * D); w.wq = take((size_t)d.H * D);
  w.lnstat = take(4); w.dw = take((size_t)d.H * D); w.du = take((size_t)d.H * D); w.dqh = take(D); w.dqn = take(D);
  w.Wvs = take(D * D); w.bo = take(D); w.dWvs = take(D * D); w.dbo = take(D);
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.H);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  w.pool_total = off;
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[CAE_NT];
    w.opt_ws_bytes = optim_workspace_bytes(cae_offsets(d, offs), CAE_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int cae_check(const ep_cae_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.H > 0, EP_E_ARG, "cae dims must be positive");
  EP_REQUIRE(d.D % d.H == 0 && (d.D / d.H) % 4 == 0 && d.D % 4 == 0, EP_E_SHAPE, "cae: D %% H == 0 and D/H, D multiples of 4");
  EP_REQUIRE(d.H <= 32 && (size_t)(2 * d.D + 256) * 4 <= 60000, EP_E_UNSUPPORTED, "cae: heads > 32 or D too large");
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "cae head: C must be positive");
  return 0;
}

static int cae_params_ok(const ep_cae_params* p, const char* what) {
  EP_REQUIRE(p && p->query && p->nq_w && p->nq_b && p->nk_w && p->nk_b && p->nv_w && p->nv_b && p->n2_w && p->n2_b && p->q_w &&
             p->k_w && p->v_w && p->proj_w && p->proj_b, EP_E_ARG, "%s: null tensor", what);
  const float* ts[] = {p->query, p->nq_w, p->nq_b, p->nk_w, p->nk_b, p->nv_w, p->nv_b, p->n2_w, p->n2_b, p->q_w, p->k_w, p->v_w,
                       p->proj_w, p->proj_b};
  for (const float* t : ts) EP_REQUIRE(aligned16(t), EP_E_ALIGN, "%s: tensors must be 16-byte aligned", what);
  return 0;
}

static PoolParams cae_pool_params(const ep_cae_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                                  const float* tokstat, const CaeWs& w) {
  PoolParams p = pool_params(x, bstride, d.B, d.N, d.D, d.H, 1.0f, x_dtype);
  p.cls = w.wq; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = index; p.tokstat = tokstat;
  return p;
}

static GemmParams cg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

// tokstat: per-token {mean, rstd} of the caller (a resident store computes them once), or nullptr: computed here
static int cae_forward_core(const ep_cae_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                            const float* tokstat, float ln_eps, const ep_cae_params& pr, const CaeWs& w, float* y,
                            hipStream_t st) {
  const int D = d.D, dh = D / d.H;
  const float scale = (float)pow((double)dh, -0.5);                        // cae_att.py:29
  if (!tokstat) {
    EP_REQUIRE(!index, EP_E_ARG, "cae: an indexed token store needs precomputed token statistics");
    EP_TRY(token_stats(x, x_dtype == EP_DTYPE_BF16, bstride, d.B, d.N, D, ln_eps, w.stats, st));
    tokstat = w.stats;
  }
  hipLaunchKernelGGL(ep_coca_q_kernel, dim3((D + 3) / 4), dim3(256), (size_t)D * 4, st, pr.query, pr.nq_w, pr.nq_b, pr.q_w, D, D,
                     ln_eps, scale, w.xhat, w.qn, w.lnstat, w.qh);
  hipLaunchKernelGGL(ep_cae_w_kernel, dim3((D + 31) / 32, d.H), dim3(1024), 0, st, w.qh, pr.k_w, pr.nk_w, D, dh, w.u, w.wq);
  hipLaunchKernelGGL(ep_cae_wv_kernel, dim3((D + 3) / 4), dim3(256), 0, st, pr.v_w, pr.nv_w, pr.nv_b, D, w.Wvs, w.bo,
                     (const float*)nullptr);
  EP_LAUNCH_CHECK("ep_cae query kernels");
  EP_TRY(pool_forward(cae_pool_params(d, x, x_dtype, bstride, index, tokstat, w), st));
  {
    GemmParams g = cg(w.P, (int64_t)d.H * D, w.Wvs, D, w.o, D, d.B, dh, D);          // o = Phat (Wv gv)_h^T + (Wv bv)_h
    g.sAz = D; g.sBz = (int64_t)dh * D; g.sCz = dh; g.bias = w.bo; g.sBiasz = dh;
    EP_TRY(gemm(true, true, g, d.H, st));
  }
  GemmParams g = cg(w.o, D, pr.proj_w, D, y, D, d.B, D, D); g.bias = pr.proj_b;
  return gemm(true, true, g, 1, st);
}

static int cae_backward_core(const ep_cae_dims& d, const void* x, int x_dtype, int64_t bstride, const int32_t* index,
                             const float* tokstat, float ln_eps, const ep_cae_params& pr, const float* dy,
                             const ep_cae_params& gr, int acc, const CaeWs& w, SideTasks sd, hipStream_t st, hipStream_t aux) {
  const int D = d.D, dh = D / d.H, B = d.B;
  const float scale = (float)pow((double)dh, -0.5);
  if (!tokstat) tokstat = w.stats;                                                     // left there by the forward
  // the weight-gradient contractions: on the aux stream, each as early as its operands exist (AuxSide, ep_internal.h)
  AuxSide ax;
  EP_TRY(aux_side_begin(ax, st, aux));
  GemmParams gWp = cg(dy, D, w.o, D, gr.proj_w, D, D, D, B); gWp.accumulate = acc; gWp.side = 1;          // dWp = dy^T o
  GemmParams gWv = cg(w.dO, D, w.P, (int64_t)d.H * D, w.dWvs, D, dh, D, B);                               // d(Wv gv)_h = dO_h^T Phat_h
  gWv.sAz = dh; gWv.extA = dh; gWv.sBz = D; gWv.extB = D; gWv.sCz = (int64_t)dh * D; gWv.side = 1;
  EP_REQUIRE(gemm_side_ok(gWp, false, false) && gemm_side_ok(gWv, false, false), EP_E_ALIGN, "cae: unaligned gradient contraction");
  side_add_gemm(sd, gWp, 1);
  EP_TRY(aux_side_fork(ax, sd));                                                        // the caller's dWc, dWp
  EP_TRY(gemm(true, false, cg(dy, D, pr.proj_w, D, w.dO, D, B, D, D), 1, st));         // dO = dy Wp
  side_add_gemm(sd, gWv, d.H);
  EP_TRY(aux_side_fork(ax, sd));
  EP_TRY(aux_side_rest(ax, sd));
  EP_TRY(colsum(dy, B, D, D, acc, gr.proj_b, st));
  EP_TRY(colsum(w.dO, B, D, D, 0, w.dbo, st));                                          // d(Wv bv)
  EP_TRY(delta_rows(w.dO, w.o, B * d.H, dh, w.ML, st, w.bo, d.H));                      // dPhat . Phat (bias taken out)
  {
    GemmParams g = cg(w.dO, D, w.Wvs, D, w.dP, (int64_t)d.H * D, B, D, dh);             // dPhat[b,h] = dO[b,h] (Wv gv)_h
    g.sAz = dh; g.sBz = (int64_t)dh * D; g.sCz = D; g.extB = D;
    EP_TRY(gemm(true, false, g, d.H, st));
  }
  PoolParams p = cae_pool_params(d, x, x_dtype, bstride, index, tokstat, w);
  p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
  EP_TRY(aux_side_before_pass(ax, sd));
  EP_TRY(pool_backward(p, w.dw, 0, st));
  EP_TRY(aux_side_join(ax));
  hipLaunchKernelGGL(ep_cae_dwv_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.dWvs, w.dbo, pr.v_w, pr.nv_w, pr.nv_b, D, acc,
                     gr.v_w, gr.nv_w, gr.nv_b, gr.n2_w, gr.n2_b);
  hipLaunchKernelGGL(ep_cae_du_kernel, dim3((D + 255) / 256), dim3(256), 0, st, w.dw, w.u, pr.nk_w, D, d.H, acc, w.du, gr.nk_w,
                     gr.nk_b);
  hipLaunchKernelGGL(ep_cae_dqh_kernel, dim3((D + 3) / 4), dim3(256), 0, st, w.du, pr.k_w, D, dh, w.dqh);
  hipLaunchKernelGGL(ep_cae_qgrad_kernel, dim3((D + 63) / 64, 16), dim3(256), 0, st, w.qh, w.dqh, w.du, w.qn, D, dh, scale, acc,
                     gr.k_w, gr.q_w);
  hipLaunchKernelGGL(ep_cae_dqn_kernel, dim3((D + 31) / 32), dim3(1024), 0, st, w.dqh, pr.q_w, D, scale, w.dqn);
  hipLaunchKernelGGL(ep_coca_lnbwd_kernel, dim3(1), dim3(256), 0, st, w.dqn, w.xhat, pr.nq_w, w.lnstat, D, acc, gr.nq_w,
                     gr.query, gr.nq_b);
  EP_LAUNCH_CHECK("ep_cae backward kernels");
  (void)ln_eps;
  return 0;
}

static ep_cae_params cae_views(float* base, const int64_t o[CAE_NT]) {
  ep_cae_params p;
  p.query = base + o[0]; p.nq_w = base + o[1]; p.nq_b = base + o[2]; p.nk_w = base + o[3]; p.nk_b = base + o[4];
  p.nv_w = base + o[5]; p.nv_b = base + o[6]; p.n2_w = base + o[7]; p.n2_b = base + o[8]; p.q_w = base + o[9];
  p.k_w = base + o[10]; p.v_w = base + o[11]; p.proj_w = base + o[12]; p.proj_b = base + o[13];
  return p;
}

}  // namespace ep

extern "C" {

size_t ep_cae_pool_workspace_bytes(const ep_cae_dims* dims) {
  if (!dims || cae_check(*dims, false) != 0) return 0;
  return cae_carve(*dims, nullptr, false).total;
}

int ep_cae_pool_forward(const ep_cae_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                        const float* token_stats_, float ln_eps, const ep_cae_params* params, float* y, void* ws,
                        size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_cae_pool_forward: null pointer");
  EP_TRY(cae_check(*dims, false));
  EP_TRY(cae_params_ok(params, "ep_cae_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_cae_pool_forward: y / ws must be 16-byte aligned");
  const CaeWs w = cae_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cae_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  return cae_forward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, ln_eps, *params, w, y, (hipStream_t)stream);
}

int ep_cae_pool_backward(const ep_cae_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                         const float* token_stats_, float ln_eps, const ep_cae_params* params, const float* dy,
                         const ep_cae_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dy && ws, EP_E_ARG, "ep_cae_pool_backward: null pointer");
  EP_TRY(cae_check(*dims, false));
  EP_TRY(cae_params_ok(params, "ep_cae_pool_backward(params)"));
  EP_TRY(cae_params_ok(grads, "ep_cae_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, dims->H));
  EP_REQUIRE(aligned16(ws) && aligned16(dy), EP_E_ALIGN, "ep_cae_pool_backward: dy / ws must be 16-byte aligned");
  const CaeWs w = cae_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cae_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return cae_backward_core(*dims, x, x_dtype, x_bstride, image_index, token_stats_, ln_eps, *params, dy, *grads, accumulate, w,
                           SideTasks{}, (hipStream_t)stream, nullptr);
}

int64_t ep_cae_head_param_offsets(const ep_cae_dims* dims, int64_t offsets[16]) { return cae_offsets(*dims, offsets); }

size_t ep_cae_head_workspace_bytes(const ep_cae_dims* dims) {
  if (!dims || cae_check(*dims, true) != 0) return 0;
  return cae_carve(*dims, nullptr, true).total;
}

int ep_cae_head_train_step(const ep_cae_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_cae_head_train_step: null pointer");
  const ep_cae_dims& d = s->dims;
  EP_TRY(cae_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const CaeWs w = cae_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cae_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CAE_NT];
  const int64_t total = cae_offsets(d, offs);
  const ep_cae_params pr = cae_views(s->params, offs), gr = cae_views(s->grads, offs);
  float* Wc = s->params + offs[14]; float* bc = s->params + offs[15];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.H));
    EP_TRY(cae_forward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, s->ln_eps, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, nullptr, nullptr, 0, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    SideTasks sd{};
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, d.D, d.C, s->grads + offs[14], s->accumulate);
    EP_REQUIRE(gemm_side_ok(gWc, false, false), EP_E_ALIGN, "cae head: unaligned classifier gradient");
    side_add_gemm(sd, gWc, 1);
    sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[15]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
    sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
    sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
    sd.total += sd.n_colsum + sd.n_stats;
    EP_TRY(cae_backward_core(d, s->x, s->x_dtype, s->x_bstride, s->image_index, s->token_stats, s->ln_eps, pr, w.dy, gr,
                             s->accumulate, w, sd, st, (hipStream_t)s->aux_stream));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    const int64_t D = d.D;
    const int64_t sizes[CAE_NT] = {D, D, D, D, D, D, D, D, D, D * D, D * D, D * D, D * D, D, (int64_t)d.C * D, d.C};
    // util/lars.py:22: trust ratio + weight decay for ndim > 1: the query token is (1, 1, D); LayerNorm vectors and biases are not
    const int trust[CAE_NT] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0};
    ep_segment segs[CAE_NT];
    for (int i = 0; i < CAE_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? CAE_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_cae_head_eval_forward(const ep_cae_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                             const float* token_stats_, float ln_eps, const float* params, const float* running_mean,
                             const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                             ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_cae_head_eval_forward: null pointer");
  const ep_cae_dims& d = *dims;
  EP_TRY(cae_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, d.H));
  const CaeWs w = cae_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cae_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CAE_NT];
  cae_offsets(d, offs);
  const ep_cae_params pr = cae_views(const_cast<float*>(params), offs);
  EP_TRY(cae_forward_core(d, x, x_dtype, x_bstride, image_index, token_stats_, ln_eps, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[14], params + offs[15], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
