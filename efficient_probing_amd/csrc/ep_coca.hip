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
__global__ __launch_bounds__(256) void ep_coca_u_kernel(const float* __restrict__ qh, const float* __restrict__ Wk,
                                                      int D, int dh, float* __restrict__ u) {
  const int d = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (d >= D) return;
  float acc = 0.f;
  for (int c = 0; c < dh; ++c) acc = fmaf(qh[h * dh + c], Wk[(int64_t)c * D + d], acc);
  u[(int64_t)h * D + d] = acc;
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
__global__ __launch_bounds__(256) void ep_coca_qgrad_kernel(const float* __restrict__ qh, const float* __restrict__ dqh,
                                                          const float* __restrict__ du, const float* __restrict__ qn,
                                                          const float* __restrict__ Wq, int D, int H, int dh, int M,
                                                          float scale, int accumulate, float* __restrict__ dWk,
                                                          float* __restrict__ dWq, float* __restrict__ dqn,
                                                          float* __restrict__ dimgq) {
  extern __shared__ float sh[];          // qh[HD] | dqh[HD] | partial[4][64]
  const int HD = H * dh;
  float* s_qh = sh; float* s_dqh = sh + HD; float* part = sh + 2 * HD;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  for (int i = tid; i < HD; i += 256) { s_qh[i] = qh[i]; s_dqh[i] = dqh[i]; }
  __syncthreads();
  const int d = blockIdx.x * 64 + tx;
  const bool ok = d < D;
  float acc = 0.f;
  if (ok) {
    for (int c = ty; c < dh; c += 4) {
      float g = 0.f;
      for (int h = 0; h < H; ++h) g = fmaf(s_qh[h * dh + c], du[(int64_t)h * D + d], g);
      float* o = dWk + (int64_t)c * D + d;
      *o = accumulate ? *o + g : g;
    }
    const float qd = qn[d] * scale;
    for (int j = ty; j < HD; j += 4) {
      acc = fmaf(Wq[(int64_t)j * D + d], s_dqh[j], acc);
      float* o = dWq + (int64_t)j * D + d;
      const float g = s_dqh[j] * qd;
      *o = accumulate ? *o + g : g;
    }
    if (!accumulate)
      for (int r = 1 + ty; r < M; r += 4) dimgq[(int64_t)r * D + d] = 0.f;
  }
  part[ty * 64 + tx] = acc;
  __syncthreads();
  if (ty == 0 && ok) dqn[d] = ((part[tx] + part[64 + tx]) + (part[128 + tx] + part[192 + tx])) * scale;
}

// LayerNorm backward of the single live row: dgamma (+)= dqn * xhat; d img_queries[0] (+)= the usual
// rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dqn * gamma.
__global__ __launch_bounds__(256) void ep_coca_lnbwd_kernel(const float* __restrict__ dqn, const float* __restrict__ xhat,
                                                          const float* __restrict__ gamma, const float* __restrict__ lnstat,
                                                          int D, int accumulate, float* __restrict__ dgamma,
                                                          float* __restrict__ dimgq0) {
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
  hipLaunchKernelGGL(ep_coca_u_kernel, dim3((D + 255) / 256, d.H), dim3(256), 0, st, w.qh, pr.to_kv, D, d.dh, w.u);
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
  } else {
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t side = aux ? aux : st;
    if (side != st) {
      EP_TRY(get_events(ev, 2));
      EP_HIP(hipEventRecord(ev[0], st));
      EP_HIP(hipStreamWaitEvent(side, ev[0], 0));
    }
    EP_TRY(side_run_standalone(sd, side));
    EP_TRY(pool_backward(p, w.du, 0, st));
    if (side != st) {
      EP_HIP(hipEventRecord(ev[1], side));
      EP_HIP(hipStreamWaitEvent(st, ev[1], 0));
    }
  }
  // dWv = sum over heads of the partials -> rows dh..2dh-1 of d to_kv
  EP_TRY(reduce_partials(w.dWvp, d.H, d.dh * D, 1.0f, accumulate, gr.to_kv + (int64_t)d.dh * D, nullptr, st));
  hipLaunchKernelGGL(ep_coca_dqh_kernel, dim3((HD + 3) / 4), dim3(256), 0, st, w.du, pr.to_kv, D, d.dh, HD, w.dqh);
  hipLaunchKernelGGL(ep_coca_qgrad_kernel, dim3((D + 63) / 64), dim3(256), (size_t)(2 * HD + 256) * 4, st, w.qh, w.dqh,
                     w.du, w.qn, pr.to_q, D, d.H, d.dh, d.M, scale, accumulate, gr.to_kv, gr.to_q, w.dqn,
                     gr.img_queries);
  hipLaunchKernelGGL(ep_coca_lnbwd_kernel, dim3(1), dim3(256), 0, st, w.dqn, w.xhat, pr.gamma, w.lnstat, D, accumulate,
                     gr.gamma, gr.img_queries);
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
