// extern "C" entry points of libep_hip.so (declared in include/ep_hip.h).
#include <stdarg.h>
#include <string.h>
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

static thread_local char g_err[512] = "";
// diagnostics (ep_debug_set_pass_events): events recorded around the two token passes of the next EP head train steps
static thread_local hipEvent_t g_pass_ev[4] = {nullptr, nullptr, nullptr, nullptr};
static inline void mark_pass(int i, hipStream_t st) { if (g_pass_ev[i]) (void)hipEventRecord(g_pass_ev[i], st); }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int cu_count() {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  cached = n;
  return n;
}

// Fork/join events for the optional aux stream: a small per-thread pool created on first use and
// reused (events carry no data; the library otherwise keeps no state).
int get_events(hipEvent_t* out, int n) {
  constexpr int MAX_DEV = 16;          // events belong to the device that was current when they were created
  static thread_local hipEvent_t pool[MAX_DEV][8] = {};
  int dev = 0;
  EP_HIP(hipGetDevice(&dev));
  EP_REQUIRE(dev >= 0 && dev < MAX_DEV && n <= 8, EP_E_ARG, "get_events: device %d / %d events not supported", dev, n);
  for (int i = 0; i < n; ++i) {
    if (!pool[dev][i]) EP_HIP(hipEventCreateWithFlags(&pool[dev][i], hipEventDisableTiming));
    out[i] = pool[dev][i];
  }
  return 0;
}

int check_tokens(const void* x, int x_dtype, int64_t x_bstride, int B, int N, int D, int Q) {
  EP_REQUIRE(x != nullptr, EP_E_ARG, "x is null");
  EP_REQUIRE(B > 0 && N > 0 && D > 0 && Q > 0, EP_E_ARG, "B, N, D, Q must be positive (got %d %d %d %d)", B, N, D, Q);
  EP_REQUIRE(x_dtype == EP_DTYPE_F32 || x_dtype == EP_DTYPE_BF16, EP_E_UNSUPPORTED, "token dtype %d not implemented (fp32 / bf16)", x_dtype);
  EP_REQUIRE(D % 4 == 0, EP_E_SHAPE, "D = %d must be a multiple of 4", D);
  EP_REQUIRE(x_dtype == EP_DTYPE_F32 || (D % 8 == 0 && x_bstride % 8 == 0), EP_E_SHAPE, "bf16 tokens: D and the batch stride must be multiples of 8");
  EP_REQUIRE(x_bstride >= (int64_t)N * D, EP_E_SHAPE, "batch stride %lld smaller than N*D", (long long)x_bstride);
  EP_REQUIRE(aligned16(x) && x_bstride % 4 == 0, EP_E_ALIGN, "token buffer / batch stride must be 16-byte aligned");
  return 0;
}

// A second side queue of the library's own (created on first use, one per device and host thread, like the event pool): the
// 32-query step has two weight-gradient contractions of ~50 us each that feed nothing before the optimizer; on ONE side stream
// they are serialised (dWc, then dWv) and the second one runs into the second token pass, whose persistent workgroups starve
// it (measured: dWv 307 us instead of 50, the step waits 40 us for it -- profiles/r05).  With dWv on its own queue both start
// as soon as their operands exist and are done when the pass starts.
static int get_side2_stream(hipStream_t* out) {
  constexpr int MAX_DEV = 16;
  static thread_local hipStream_t pool[MAX_DEV] = {};
  int dev = 0;
  EP_HIP(hipGetDevice(&dev));
  EP_REQUIRE(dev >= 0 && dev < MAX_DEV, EP_E_ARG, "get_side2_stream: device %d not supported", dev);
  if (!pool[dev]) EP_HIP(hipStreamCreateWithFlags(&pool[dev], hipStreamNonBlocking));
  *out = pool[dev];
  return 0;
}

// forward entry points also take fp16-STORED tokens (EP_DTYPE_F16, ABI v24): same layout rules as bf16
static int check_tokens_fwd(const void* x, int x_dtype, int64_t x_bstride, int B, int N, int D, int Q) {
  return check_tokens(x, x_dtype == EP_DTYPE_F16 ? EP_DTYPE_BF16 : x_dtype, x_bstride, B, N, D, Q);
}

bool project_dp_thin_ok(int D, int Dp, int Q);           // ep_tail.hip: thin query slices (Dq <= 32)
int project_dp_thin(const float* dy, const float* Wv, int B, int D, int Dp, int Q, float* dP, hipStream_t st);
bool project_dp_slice_ok(const float* dy, const float* Wv, const float* dP, int D, int Dp, int Q);   // ep_dp_slice.hip: the same on the bf16 matrix cores (round 6)
int project_dp_slice(const float* dy, const float* Wv, int B, int D, int Dp, int Q, float* dP, hipStream_t st, const float* yv, float* ML);

struct HeadWs {
  float *P, *S, *ML, *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy, *dP;
  float* ypart;                                // in-pass value projection: IP_YPARTS K-quarter partials of y (ep_inpass.h)
  float* colstat;                              // per-32-row-tile column sums of dz and dz z (folded BatchNorm backward)
  int *ycnt, *dcnt, *tick, *iperr; int nrb;    // arrival counters per 32-image row block and the image ticket counter
                                               // (each on a 128-byte line of its own, zero between steps), give-up count
  uint16_t *plWv, *plWvT, *plWc, *plWcT;       // bf16 planes of the two weight matrices, both orientations (ep_planes.hip)
  uint16_t *ptP, *ptZ; float *dyT, *dlT;       // weight gradients on the planes kernel: planes of P^T and z^T, fp32 dy^T and dlogits^T
  float *wgks_c, *wgks_v; size_t wgks_floats;  // K-slice scratch of the two weight gradients where they run on side queues (ep_gemm.hip)
  void* pool_ws; size_t pool_ws_bytes;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

// Contractions against pre-split weight planes on the bf16 matrix cores (ep_planes.hip) -- EP_GEMM_PLANES: 0 off, 1 all four
// critical-path contractions (needs the per-query slice width to be a multiple of the MFMA K (32) for the dP contraction;
// switches the in-pass contractions off), 2 the classifier's two (logits = z Wc^T, dz = dlogits Wc; the planes of Wc are
// split beside the first token pass).  Unset: mode 1 for D >= 2048, off below.  Measured on MI355X, ms per step, f32 kernels
// against mode 1 (mode 2): 196 x 4096 (DINOv3 ViT-7B) 2.67 -> 2.49; 256 x 1152 0.746 -> 0.749; 196 x 1024 0.521 -> 0.553;
// 256 x 768 0.437 -> 0.460 (0.442).  Kernels alone at 256 x 768: logits 14.6 against 20.4 us, dz 16.4 against 20.8 us; at
// 1024 x 4096 x 4096 (the value projection of the 7B tokens) 184 against 353 us -- the small shapes lose it again to the split
// launch beside the HBM-bound first pass, the half-empty 96-column tiles of their per-query projection and the in-pass dP.
static int head_planes_mode(const ep_head_dims& d) {
  static int on = -2;
  if (on == -2) { const char* e = getenv("EP_GEMM_PLANES"); on = e ? atoi(e) : -1; }
  const int Dp = d.D / d.d_out;
  // unset: all four contractions for wide rows; below, the classifier's two -- their planes come for free since round 4 (the
  // optimizer's update kernel writes them, ep_optim.hip: tile_update_emit; ep_head_step.planes_valid): 256 x 768, same box,
  // mode 0 against mode 2: 0.4343 -> 0.4285 ms (f32 tokens), 0.3141 -> 0.3060 ms (bf16-stored tokens)
  // the AMP-bf16 arithmetic mode (ep_head_step.arith, gemm_arith()) runs all four against the planes whatever the width
  const int mode = gemm_arith() == 1 ? 1 : on >= 0 ? on : (d.D >= 2048 ? 1 : 2);
  if (!mode || d.D % 4 != 0 || Dp % 4 != 0) return 0;
  // mode 1 at any slice width since round 6: only dP = dy_q Wv_q contracts over a query's SLICE of the planes' permuted k-order
  // (it needs the slice to start and end on a group of 32: head_planes_dp) -- elsewhere it falls back to the thin-slice /
  // exact-f32 kernels while y, logits, dz (and in the AMP-bf16 mode the single-product weight gradients) stay on the planes.
  // The fp32 mode keeps the round-5 choice (no planes at all) for such widths unless asked by EP_GEMM_PLANES=1.
  if (mode == 1) return ((Dp / d.Q) % 32 == 0 || gemm_arith() == 1 || on == 1) ? 1 : 0;
  return 2;
}
static bool head_planes_dp(const ep_head_dims& d) { return ((d.D / d.d_out) / d.Q) % 32 == 0; }
static bool head_planes_ok(const ep_head_dims& d) { return head_planes_mode(d) != 0; }
// The weight gradients dWv = dy^T P and dWc = dlogits^T z (sums over the batch index) on the planes kernel as well
// (EP_PLANES_WGRAD, default on in mode 1): the activation that plays the weight -- P, z -- is split into planes of its
// transpose (ep_planes_split_kernel's transposed orientation), the other one transposed in fp32.  196 x 4096 tokens: the split
// of P (134 MB read, 201 MB written) runs on the side stream beside the contractions between the passes.
static bool head_planes_wgrad(const ep_head_dims& d) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_PLANES_WGRAD"); on = e ? atoi(e) : 1; }
  // round 4: with the bf16 x3 weight-gradient tile (ep_wgrad3.h, EP_GEMM_B3) both operands are split on the fly in their
  // natural T / T layout -- no transposes, no planes of P^T -- so this path is only taken when that tile is switched off
  // or asked for explicitly (EP_PLANES_WGRAD=2)
  if (on != 2 && gemm_b3_on()) return false;
  return on && head_planes_mode(d) == 1 && d.B % 4 == 0 && d.B >= 128;
}
static HeadWs carve(const ep_head_dims& d, void* base) {
  HeadWs w{};
  const int Dp = d.D / d.d_out;
  w.ldl = (d.C + 3) / 4 * 4;
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B;
  w.P = take(B * d.Q * d.D);
  w.S = take(B * d.Q * d.N);
  w.ML = take(B * d.Q * 4);
  w.y = take(B * Dp);
  w.z = take(B * Dp);
  w.rstd = take(Dp);
  w.logits = take(B * w.ldl);
  w.dlogits = take(B * w.ldl);
  w.rowstat = take(B * 4);
  w.bnpart = take(bn_workspace_bytes(d.B, Dp) / sizeof(float));
  w.dz = take(B * Dp);
  w.dy = take(B * Dp);
  w.dP = take(B * d.Q * d.D);
  w.ypart = take((size_t)IP_YPARTS * B * Dp);
  w.colstat = take(((size_t)(d.B + 31) / 32 + 1) * 2 * Dp);
  w.nrb = (d.B + 31) / 32;
  // one counter per 128-byte line (IP_CNT_STRIDE ints apart): ycnt | dcnt | ticket | give-up count
  w.ycnt = reinterpret_cast<int*>(take(2 * (size_t)w.nrb * 32 + 64));
  w.dcnt = w.ycnt ? w.ycnt + (size_t)w.nrb * 32 : nullptr;
  w.tick = w.ycnt ? w.ycnt + 2 * (size_t)w.nrb * 32 : nullptr;
  w.iperr = w.ycnt ? w.ycnt + 2 * (size_t)w.nrb * 32 + 32 : nullptr;
  w.pool_ws_bytes = pool_workspace_bytes(d.B, d.N, d.D, d.Q);
  w.pool_ws = take(w.pool_ws_bytes / sizeof(float));
  int64_t offs[4];
  const int64_t total = ep_head_param_offsets(&d, offs);
  w.opt_ws_bytes = optim_workspace_bytes(total, 4);
  w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  if (head_planes_ok(d)) {            // the bf16 weight planes exist only when the (opt-in) planes contractions will run
    auto take16 = [&](size_t n) { return reinterpret_cast<uint16_t*>(take((n + 1) / 2)); };
    w.plWv = take16(planes_elems(Dp, d.D)); w.plWvT = take16(planes_elems(d.D, Dp));
    w.plWc = take16(planes_elems(d.C, Dp)); w.plWcT = take16(planes_elems(Dp, d.C));
    if (head_planes_wgrad(d)) {
      w.ptP = take16(planes_elems(d.Q * d.D, d.B)); w.ptZ = take16(planes_elems(Dp, d.B));
      w.dyT = take((size_t)Dp * d.B); w.dlT = take((size_t)d.C * d.B);
    }
  }
  {
    // K-slice scratch (GemmParams.skws) for dWc / dWv as launches of their own: only where a gradient is fewer than three 32-row
    // tiles per CU -- up to 4 slices each
    const size_t mnc = (size_t)d.C * Dp, mnv = (size_t)Dp * d.D;
    const size_t tiles_c = (size_t)((Dp + 63) / 64) * ((d.C + 31) / 32), tiles_v = (size_t)((d.D + 63) / 64) * ((Dp / d.Q + 31) / 32) * d.Q;
    static int ks_on = -1;                            // (EP_B3_SPLITK_WGS, off by default: ep_gemm.hip)
    if (ks_on < 0) { const char* e = getenv("EP_B3_SPLITK_WGS"); ks_on = e ? atoi(e) : 0; }
    const size_t need = (ks_on > 0 && (tiles_c < 768 || tiles_v < 768)) ? 4 * (mnc > mnv ? mnc : mnv) : 0;
    w.wgks_floats = need;
    w.wgks_c = need ? take(need) : nullptr;
    w.wgks_v = need ? take(need) : nullptr;
  }
  w.total = off;
  return w;
}

static int head_planes_split(const ep_head_dims& d, const HeadWs& w, const float* Wv, const float* Wc, hipStream_t st) {
  const int Dp = d.D / d.d_out;
  PlaneSpec sp[2] = {{Wc, d.C, Dp, Dp, w.plWc, w.plWcT}, {Wv, Dp, d.D, d.D, w.plWv, w.plWvT}};
  return planes_split(sp, head_planes_mode(d) == 1 ? 2 : 1, st);
}
static GemmParams planes_gemm(const float* A, int64_t lda, int64_t sAz, const uint16_t* pl, int rowsW, int Kw, int64_t sBpz,
                              float* C, int64_t ldc, int64_t sCz, int M, int N, int K, const float* bias) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.sAz = sAz; g.C = C; g.ldc = ldc; g.sCz = sCz; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.bias = bias;
  g.Bpl = pl; g.ldbp = (int64_t)round_up((size_t)Kw, 32); g.pl_term = (int64_t)rowsW * g.ldbp; g.sBpz = sBpz;
  return g;
}
// y[b, q Dq + c] = P[b, q, :] . Wv[q Dq + c, :]
static int project_forward_pl(const HeadWs& w, const ep_head_dims& d, hipStream_t st) {
  const int Dp = d.D / d.d_out, Dq = Dp / d.Q;
  const int64_t ld = (int64_t)round_up((size_t)d.D, 32);
  return gemm_planes(planes_gemm(w.P, (int64_t)d.Q * d.D, d.D, w.plWv, Dp, d.D, (int64_t)Dq * ld, w.y, Dp, Dq, d.B, Dq, d.D, nullptr), d.Q, st);
}
// dP[b, q, :] = dy[b, q Dq : (q+1) Dq] . Wv[q Dq : (q+1) Dq, :]   (planes of Wv^T, contraction offset q Dq)
static int project_backward_dP_pl(const HeadWs& w, const ep_head_dims& d, hipStream_t st) {
  const int Dp = d.D / d.d_out, Dq = Dp / d.Q;
  return gemm_planes(planes_gemm(w.dy, Dp, Dq, w.plWvT, d.D, Dp, Dq, w.dP, (int64_t)d.Q * d.D, d.D, d.B, d.D, Dq, nullptr), d.Q, st);
}
static int linear_forward_pl(const HeadWs& w, const ep_head_dims& d, const float* bc, hipStream_t st) {
  const int Dp = d.D / d.d_out;
  return gemm_planes(planes_gemm(w.z, Dp, 0, w.plWc, d.C, Dp, 0, w.logits, w.ldl, 0, d.B, d.C, Dp, bc), 1, st);
}
static int linear_backward_dz_pl(const HeadWs& w, const ep_head_dims& d, hipStream_t st) {
  const int Dp = d.D / d.d_out;
  return gemm_planes(planes_gemm(w.dlogits, w.ldl, 0, w.plWcT, Dp, d.C, 0, w.dz, Dp, 0, d.B, Dp, d.C, nullptr), 1, st);
}

// dWc[c, :] (+)= sum_b dlogits[b, c] z[b, :]  and  dWv[q Dq + m, :] (+)= sum_b dy[b, q Dq + m] P[b, q, :]  on the planes kernel
static int wgrad_dwc_pl(const HeadWs& w, const ep_head_dims& d, float* dWc, int accumulate, hipStream_t st) {
  const int Dp = d.D / d.d_out;
  EP_TRY(transpose_f32(w.dlogits, d.B, d.C, w.ldl, w.dlT, d.B, st));
  PlaneSpec sp{w.z, d.B, Dp, Dp, nullptr, w.ptZ};
  EP_TRY(planes_split(&sp, 1, st));
  GemmParams g = planes_gemm(w.dlT, d.B, 0, w.ptZ, Dp, d.B, 0, dWc, Dp, 0, d.C, Dp, d.B, nullptr);
  g.accumulate = accumulate;
  return gemm_planes(g, 1, st);
}
static int wgrad_split_P(const HeadWs& w, const ep_head_dims& d, hipStream_t st) {
  PlaneSpec sp{w.P, d.B, d.Q * d.D, (int64_t)d.Q * d.D, nullptr, w.ptP};
  return planes_split(&sp, 1, st);
}
static int wgrad_dwv_pl(const HeadWs& w, const ep_head_dims& d, float* dWv, int accumulate, hipStream_t st) {
  const int Dp = d.D / d.d_out, Dq = Dp / d.Q;
  const int64_t ldb = (int64_t)round_up((size_t)d.B, 32);
  EP_TRY(transpose_f32(w.dy, d.B, Dp, Dp, w.dyT, d.B, st));
  GemmParams g = planes_gemm(w.dyT, d.B, (int64_t)Dq * d.B, w.ptP, d.Q * d.D, d.B, (int64_t)d.D * ldb, dWv, d.D, (int64_t)Dq * d.D,
                             Dq, d.D, d.B, nullptr);
  g.accumulate = accumulate;
  return gemm_planes(g, d.Q, st);
}

static int check_dims(const ep_head_dims& d) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.Q > 0 && d.C > 0 && d.d_out > 0, EP_E_ARG, "head dims must be positive");
  EP_REQUIRE(d.D % (d.d_out * d.Q) == 0, EP_E_SHAPE, "D %% (d_out*Q) != 0 (D=%d d_out=%d Q=%d) -- same constraint as reference ep.py:40", d.D, d.d_out, d.Q);
  EP_REQUIRE((d.D / d.d_out) % 4 == 0 && d.D % 4 == 0, EP_E_SHAPE, "D and D/d_out must be multiples of 4");
  return 0;
}

PoolParams pool_params(const void* x, int64_t x_bstride, int B, int N, int D, int Q, float scale, int x_dtype) {
  PoolParams p{};
  p.x = static_cast<const float*>(x); p.x_bf16 = x_dtype == EP_DTYPE_BF16 ? 1 : (x_dtype == EP_DTYPE_F16 ? 2 : 0); p.x_bstride = x_bstride; p.B = B; p.N = N; p.D = D; p.Q = Q; p.scale = scale;
  return p;
}

static int project_forward(const float* P, const float* Wv, int B, int D, int Dp, int Q, float* y, hipStream_t st) {
  const int Dq = Dp / Q;
  GemmParams g{};
  g.A = P; g.lda = (int64_t)Q * D; g.sAz = D;
  g.B = Wv; g.ldb = D; g.sBz = (int64_t)Dq * D;
  g.C = y; g.ldc = Dp; g.sCz = Dq;
  g.M = B; g.N = Dq; g.K = D; g.alpha = 1.f; g.extA = D; g.extB = D;
  return gemm(true, true, g, Q, st);
}

// The `side` hint of the two weight-gradient contractions (GemmParams.side: "runs beside a vector-issue-bound token pass: keep to
// the exact-f32 tile below 8 GFLOP", ep_gemm.hip).  The 32-query step runs them beside the small kernels between the passes and
// the matrix-pipe-bound 32-query passes; there the bf16 x3 tile measured faster (0.7225 -> 0.714 ms fp32 tokens, 0.440 -> 0.435 ms
// bf16, round 5), so that branch clears the hint for its calls.
static thread_local int t_wgrad_side = 1;
// K-slice scratch of the two weight gradients (set for the duration of the side-queue branch of ep_head_train_step): region 0 = dWc,
// 1 = dWv -- they run concurrently on two queues
static thread_local float* t_wgks[2] = {nullptr, nullptr};
static thread_local size_t t_wgks_floats = 0;
// dWv[q*Dq + c, d] (+)= sum_b dy[b, q*Dq + c] * P[b, q, d]   (Q batched T/T contractions over the batch)
static GemmParams dwv_gemm(const float* dy, const float* P, int B, int D, int Dp, int Q, float* dWv, int accumulate) {
  const int Dq = Dp / Q;
  GemmParams g{};
  g.A = dy; g.lda = Dp; g.sAz = Dq; g.extA = Dq;
  g.B = P; g.ldb = (int64_t)Q * D; g.sBz = D; g.extB = D;
  g.C = dWv; g.ldc = D; g.sCz = (int64_t)Dq * D;
  g.M = Dq; g.N = D; g.K = B; g.alpha = 1.f; g.accumulate = accumulate; g.side = t_wgrad_side;
  g.nterms = gemm_arith() == 1 ? 1 : 0;            // (rides into the side tiles of the second token pass as well)
  g.skws = t_wgks[1]; g.skws_floats = t_wgks[1] ? t_wgks_floats : 0;
  return g;
}
// dWc[c, k] (+)= sum_b dlogits[b, c] * z[b, k]
GemmParams dwc_gemm(const float* dl, int ldl, const float* z, int B, int Dp, int C, float* dWc, int accumulate) {
  GemmParams g{};
  g.A = dl; g.lda = ldl; g.extA = ldl; g.B = z; g.ldb = Dp; g.extB = Dp; g.C = dWc; g.ldc = Dp;
  g.M = C; g.N = Dp; g.K = B; g.alpha = 1.f; g.accumulate = accumulate; g.side = t_wgrad_side;
  g.nterms = gemm_arith() == 1 ? 1 : 0;
  g.skws = t_wgks[0]; g.skws_floats = t_wgks[0] ? t_wgks_floats : 0;
  return g;
}

static int project_backward(const float* dy, const float* y, const float* P, const float* Wv, int B, int D, int Dp,
                            int Q, float* dP, float* dWv, float* ML, int accumulate, hipStream_t st) {
  const int Dq = Dp / Q;
  const bool slice = dP && project_dp_slice_ok(dy, Wv, dP, D, Dp, Q);
  const bool slice_delta = slice && ML && y && aligned16(y) && aligned16(ML);       // the delta rows inside the dP kernel (one launch less)
  if (ML && y && !slice_delta) EP_TRY(delta_rows(dy, y, B * Q, Dq, ML, st));
  if (slice) {
    EP_TRY(project_dp_slice(dy, Wv, B, D, Dp, Q, dP, st, slice_delta ? y : nullptr, slice_delta ? ML : nullptr));
  } else if (dP && project_dp_thin_ok(D, Dp, Q)) {
    EP_TRY(project_dp_thin(dy, Wv, B, D, Dp, Q, dP, st));
  } else if (dP) {
    GemmParams g{};
    g.A = dy; g.lda = Dp; g.sAz = Dq;
    g.B = Wv; g.ldb = D; g.sBz = (int64_t)Dq * D; g.extB = D;
    g.C = dP; g.ldc = (int64_t)Q * D; g.sCz = D;
    g.M = B; g.N = D; g.K = Dq; g.alpha = 1.f;
    EP_TRY(gemm(true, false, g, Q, st));
  }
  if (dWv) EP_TRY(gemm(false, false, dwv_gemm(dy, P, B, D, Dp, Q, dWv, accumulate), Q, st));
  return 0;
}

int linear_forward(const float* z, const float* Wc, const float* bc, int B, int Dp, int C, float* logits,
                          int ldl, hipStream_t st) {
  GemmParams g{};
  g.A = z; g.lda = Dp; g.B = Wc; g.ldb = Dp; g.C = logits; g.ldc = ldl; g.bias = bc;
  g.M = B; g.N = C; g.K = Dp; g.alpha = 1.f;
  return gemm(true, true, g, 1, st);
}

int linear_backward(const float* dl, int ldl, const float* z, const float* Wc, int B, int Dp, int C,
                           float* dz, float* dWc, float* dbc, int accumulate, hipStream_t st) {
  if (dz) {
    GemmParams g{};
    g.A = dl; g.lda = ldl; g.B = Wc; g.ldb = Dp; g.extB = Dp; g.C = dz; g.ldc = Dp;
    g.M = B; g.N = Dp; g.K = C; g.alpha = 1.f;
    EP_TRY(gemm(true, false, g, 1, st));
  }
  if (dWc) EP_TRY(gemm(false, false, dwc_gemm(dl, ldl, z, B, Dp, C, dWc, accumulate), 1, st));
  if (dbc) EP_TRY(colsum(dl, B, C, ldl, accumulate, dbc, st));
  return 0;
}

void side_add_gemm(SideTasks& sd, const GemmParams& g, int batch) {
  const int i = sd.n_gemm++;
  sd.g[i] = g;
  // tile height: 64 rows where they divide (dWc: 1000 x 768), 32 where 64 would leave a third of the tiles half empty
  // (dWv: 96 rows per query).  Same-box A/B of the whole step (EP_SIDE_BM, diagnostics): this mix 0.4465 ms, all 32-row
  // tiles 0.4535 ms, all 64-row tiles 0.459 ms.
  static int force_bm = -1;
  if (force_bm < 0) { const char* e = getenv("EP_SIDE_BM"); force_bm = e ? atoi(e) : 0; }
  sd.bm[i] = (g.M % 64 == 0 || g.M >= 256) ? 64 : 32;
  if (force_bm == 32 || force_bm == 64) sd.bm[i] = force_bm;
  sd.gx[i] = (g.N + 63) / 64; sd.gy[i] = (g.M + sd.bm[i] - 1) / sd.bm[i]; sd.gz[i] = batch;
  static int xcd = -1;
  if (xcd < 0) { const char* e = getenv("EP_SIDE_XCD"); xcd = e ? atoi(e) : 1; }
  sd.xcd_order = xcd;
  sd.b3 = gemm_b3_on() && g.K >= 64 ? (i == 0 ? 1 : sd.b3) : 0;      // (one switch per launch: all of its contractions or none)
  sd.total += sd.gx[i] * sd.gy[i] * sd.gz[i];
}

bool side_add_colsum(SideTasks& sd, const float* src, int B, int ncol, int ld, int accumulate, float* out) {
  if (sd.n_xcs >= 4) return false;
  const int i = sd.n_xcs++;
  sd.xcs_src[i] = src; sd.xcs_out[i] = out; sd.xcs_B[i] = B; sd.xcs_ncol[i] = ncol; sd.xcs_ld[i] = ld; sd.xcs_acc[i] = accumulate;
  sd.xcs_blocks[i] = (ncol + 15) / 16;
  sd.total += sd.xcs_blocks[i];
  return true;
}

int side_run_standalone(const SideTasks& sd, hipStream_t st) {
  for (int i = 0; i < sd.n_gemm; ++i) EP_TRY(gemm(false, false, sd.g[i], sd.gz[i], st));
  if (sd.n_colsum) EP_TRY(colsum(sd.cs_src, sd.cs_B, sd.cs_ncol, sd.cs_ld, sd.cs_accumulate, sd.cs_out, st));
  for (int i = 0; i < sd.n_xcs; ++i) EP_TRY(colsum(sd.xcs_src[i], sd.xcs_B[i], sd.xcs_ncol[i], sd.xcs_ld[i], sd.xcs_acc[i], sd.xcs_out[i], st));
  if (sd.n_stats) EP_TRY(ce_stats(sd.rowstat, sd.rs_B, sd.stats, st));
  return 0;
}

int aux_side_begin(AuxSide& a, hipStream_t st, hipStream_t aux, bool two) {
  static int early_env = -1;
  if (early_env < 0) { const char* e = getenv("EP_WGRAD_EARLY"); early_env = e ? atoi(e) : 1; }
  static int two_env = -1;                            // EP_AUX_TWO=0: one side queue for every head
  if (two_env < 0) { const char* e = getenv("EP_AUX_TWO"); two_env = e ? atoi(e) : 1; }
  a = AuxSide{};
  a.st = st; a.side = aux ? aux : st;
  a.early = early_env && a.side != st;
  if (a.side != st) EP_TRY(get_events(a.ev, 8));
  if (two && two_env && a.early) EP_TRY(get_side2_stream(&a.side2));
  return 0;
}
static int aux_side_sync(AuxSide& a, hipStream_t target = nullptr) {   // `target` (default: the aux stream) waits for everything enqueued on `st` so far
  hipEvent_t e = a.ev[a.nev % 5];                     // (ev[5] / ev[6] are the joins'; re-recording an event whose earlier wait is already
  ++a.nev;                                            // enqueued is legal: a wait refers to the record in front of it)
  EP_HIP(hipEventRecord(e, a.st));
  EP_HIP(hipStreamWaitEvent(target ? target : a.side, e, 0));
  return 0;
}
// An early contraction runs beside the chain in front of the pass, not beside the pass: the `side` hint (which keeps a
// contraction beside a vector-issue-bound stream on the exact-f32 kernel, ep_gemm.hip: gemm_b3_ok) does not apply.  EP_AUX_B3=0 keeps it.
static GemmParams aux_early_params(const GemmParams& g) {
  static int b3 = -1;
  if (b3 < 0) { const char* e = getenv("EP_AUX_B3"); b3 = e ? atoi(e) : 1; }
  GemmParams q = g;
  if (b3) q.side = 0;
  return q;
}
int aux_side_fork(AuxSide& a, const SideTasks& sd) {
  if (!a.early || a.launched >= sd.n_gemm) return 0;
  hipStream_t target = a.side;
  if (a.side2 && (a.nfork & 1)) { target = a.side2; a.used2 = true; }
  ++a.nfork;
  EP_TRY(aux_side_sync(a, target));
  for (; a.launched < sd.n_gemm; ++a.launched) EP_TRY(gemm(false, false, aux_early_params(sd.g[a.launched]), sd.gz[a.launched], target));
  return 0;
}
int aux_side_rest(AuxSide& a, const SideTasks& sd) {
  if (!a.early || a.rest_done) return 0;
  EP_TRY(aux_side_sync(a));
  SideTasks rest = sd;
  rest.n_gemm = 0;
  EP_TRY(side_run_standalone(rest, a.side));
  a.rest_done = true;
  return 0;
}
int aux_side_before_pass(AuxSide& a, const SideTasks& sd) {
  if (a.rest_done && a.launched >= sd.n_gemm) return 0;
  if (a.side != a.st) EP_TRY(aux_side_sync(a));
  for (; a.launched < sd.n_gemm; ++a.launched) EP_TRY(gemm(false, false, sd.g[a.launched], sd.gz[a.launched], a.side));
  if (!a.rest_done) {
    SideTasks rest = sd;
    rest.n_gemm = 0;
    EP_TRY(side_run_standalone(rest, a.side));
    a.rest_done = true;
  }
  return 0;
}
int aux_side_gemm(AuxSide& a, const GemmParams& g, int batch) {
  if (!a.early) return gemm(false, false, g, batch, a.st);
  EP_TRY(aux_side_sync(a));
  return gemm(false, false, aux_early_params(g), batch, a.side);
}
int classifier_backward(AuxSide& a, const float* dlogits, int ldl, const float* z, const float* Wc, int B, int D, int C, float* dz,
                        float* dWc, float* dbc, int accumulate) {
  if (!a.early) return linear_backward(dlogits, ldl, z, Wc, B, D, C, dz, dWc, dbc, accumulate, a.st);
  GemmParams g = dwc_gemm(dlogits, ldl, z, B, D, C, dWc, accumulate);
  g.side = 1;
  EP_TRY(aux_side_gemm(a, g, 1));
  EP_TRY(colsum(dlogits, B, C, ldl, accumulate, dbc, a.side));
  return linear_backward(dlogits, ldl, z, Wc, B, D, C, dz, nullptr, nullptr, 0, a.st);
}
int aux_side_join(AuxSide& a) {
  if (a.side == a.st) return 0;
  EP_HIP(hipEventRecord(a.ev[5], a.side));
  EP_HIP(hipStreamWaitEvent(a.st, a.ev[5], 0));
  if (a.used2) {
    EP_HIP(hipEventRecord(a.ev[6], a.side2));
    EP_HIP(hipStreamWaitEvent(a.st, a.ev[6], 0));
  }
  return 0;
}

}  // namespace ep

using namespace ep;

extern "C" {

int ep_version(void) { return EP_ABI_VERSION; }
const char* ep_last_error_string(void) { return g_err; }
int ep_device_cu_count(void) { return cu_count(); }
int ep_debug_force_generic_pool(int on) { return debug_force_generic(on); }
int ep_debug_set_pass_events(void* fwd_begin, void* fwd_end, void* bwd_begin, void* bwd_end) {
  g_pass_ev[0] = (hipEvent_t)fwd_begin; g_pass_ev[1] = (hipEvent_t)fwd_end;
  g_pass_ev[2] = (hipEvent_t)bwd_begin; g_pass_ev[3] = (hipEvent_t)bwd_end;
  return 0;
}

size_t ep_pool_workspace_bytes(int B, int N, int D, int Q) { return pool_workspace_bytes(B, N, D, Q); }
const char* ep_pool_kernel_name(int B, int N, int D, int Q, int backward) { return pool_kernel_family(B, N, D, Q, backward); }
const char* ep_linear_kernel_name(int M, int N, int K) {
  GemmParams g{};
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  g.A = reinterpret_cast<const float*>(uintptr_t(1) << 20); g.B = g.A;          // (only alignment is looked at)
  return gemm_kernel_name(true, true, g, 1);
}
const char* ep_pool_kernel_name_ex(int B, int N, int D, int Q, int backward, int x_dtype) {
  return pool_kernel_family(B, N, D, Q, backward, x_dtype == EP_DTYPE_BF16 ? 1 : (x_dtype == EP_DTYPE_F16 ? 2 : 0));
}

int ep_pool_forward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                    const float* cls_token, int64_t cls_bstride, int Q, float scale, float* P, float* S, float* ML,
                    void* workspace, size_t workspace_bytes, ep_stream_t stream) {
  (void)workspace; (void)workspace_bytes;
  EP_TRY(check_tokens_fwd(x, x_dtype, x_bstride, B, N, D, Q));
  EP_REQUIRE(cls_token && P && S && ML, EP_E_ARG, "ep_pool_forward: null pointer");
  EP_REQUIRE(x_dtype != EP_DTYPE_F16 || cls_bstride == 0, EP_E_UNSUPPORTED, "fp16-stored tokens: shared query rows only");
  EP_REQUIRE(aligned16(cls_token) && aligned16(P) && aligned16(ML) && cls_bstride % 4 == 0, EP_E_ALIGN,
             "ep_pool_forward: cls_token / P / ML must be 16-byte aligned");
  PoolParams p = pool_params(x, x_bstride, B, N, D, Q, scale, x_dtype);
  p.cls = cls_token; p.cls_bstride = cls_bstride; p.P = P; p.S = S; p.ML = ML; p.index = image_index;
  return pool_forward(p, (hipStream_t)stream);
}

int ep_pool_backward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                     int Q, float scale, const float* S, const float* ML, const float* dP, float* dcls, int accumulate,
                     void* workspace, size_t workspace_bytes, ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, Q));
  EP_REQUIRE(S && ML && dP && dcls && workspace, EP_E_ARG, "ep_pool_backward: null pointer");
  EP_REQUIRE(aligned16(dP) && aligned16(dcls) && aligned16(ML) && aligned16(workspace), EP_E_ALIGN,
             "ep_pool_backward: dP / dcls / ML / workspace must be 16-byte aligned");
  EP_REQUIRE(workspace_bytes >= pool_workspace_bytes(B, N, D, Q), EP_E_WORKSPACE,
             "ep_pool_backward: workspace %zu < %zu", workspace_bytes, pool_workspace_bytes(B, N, D, Q));
  PoolParams p = pool_params(x, x_bstride, B, N, D, Q, scale, x_dtype);
  p.S = const_cast<float*>(S); p.ML = const_cast<float*>(ML); p.dP = dP; p.Gpart = static_cast<float*>(workspace);
  p.index = image_index;
  return pool_backward(p, dcls, accumulate, (hipStream_t)stream);
}

int ep_pool_backward_per_image(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N,
                               int D, int Q, float scale, const float* S, const float* ML, const float* dP, float* dq,
                               ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, Q));
  EP_REQUIRE(S && ML && dP && dq, EP_E_ARG, "ep_pool_backward_per_image: null pointer");
  EP_REQUIRE(aligned16(dP) && aligned16(dq) && aligned16(ML), EP_E_ALIGN,
             "ep_pool_backward_per_image: dP / dq / ML must be 16-byte aligned");
  PoolParams p = pool_params(x, x_bstride, B, N, D, Q, scale, x_dtype);
  p.S = const_cast<float*>(S); p.ML = const_cast<float*>(ML); p.dP = dP; p.index = image_index;
  p.cls_bstride = (int64_t)Q * D;                     // per-image query rows (selects the per-image kernels)
  return pool_backward_per_image(p, dq, (hipStream_t)stream);
}

int ep_token_stats(const void* x, int x_dtype, int64_t x_bstride, int B, int N, int D, float eps, float* stats,
                   ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, 1));
  EP_REQUIRE(stats, EP_E_ARG, "ep_token_stats: null output");
  return token_stats(x, x_dtype == EP_DTYPE_BF16, x_bstride, B, N, D, eps, stats, (hipStream_t)stream);
}

int ep_pool_forward_ln(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                       const float* cls_token, int64_t cls_bstride, int Q, float scale, const float* token_stats_,
                       float* P, float* S, float* ML, void* workspace, size_t workspace_bytes, ep_stream_t stream) {
  (void)workspace; (void)workspace_bytes;
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, Q));
  EP_REQUIRE(cls_token && P && S && ML && token_stats_, EP_E_ARG, "ep_pool_forward_ln: null pointer");
  EP_REQUIRE(aligned16(cls_token) && aligned16(P) && aligned16(ML) && cls_bstride % 4 == 0, EP_E_ALIGN,
             "ep_pool_forward_ln: cls_token / P / ML must be 16-byte aligned");
  PoolParams p = pool_params(x, x_bstride, B, N, D, Q, scale, x_dtype);
  p.cls = cls_token; p.cls_bstride = cls_bstride; p.P = P; p.S = S; p.ML = ML; p.index = image_index; p.tokstat = token_stats_;
  return pool_forward(p, (hipStream_t)stream);
}

int ep_pool_backward_ln(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                        int Q, float scale, const float* token_stats_, const float* S, const float* ML, const float* dP,
                        float* dcls, int accumulate, void* workspace, size_t workspace_bytes, ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, Q));
  EP_REQUIRE(S && ML && dP && dcls && workspace && token_stats_, EP_E_ARG, "ep_pool_backward_ln: null pointer");
  EP_REQUIRE(aligned16(dP) && aligned16(dcls) && aligned16(ML) && aligned16(workspace), EP_E_ALIGN,
             "ep_pool_backward_ln: dP / dcls / ML / workspace must be 16-byte aligned");
  EP_REQUIRE(workspace_bytes >= pool_workspace_bytes(B, N, D, Q), EP_E_WORKSPACE,
             "ep_pool_backward_ln: workspace %zu < %zu", workspace_bytes, pool_workspace_bytes(B, N, D, Q));
  PoolParams p = pool_params(x, x_bstride, B, N, D, Q, scale, x_dtype);
  p.S = const_cast<float*>(S); p.ML = const_cast<float*>(ML); p.dP = dP; p.Gpart = static_cast<float*>(workspace);
  p.index = image_index; p.tokstat = token_stats_;
  return pool_backward(p, dcls, accumulate, (hipStream_t)stream);
}

int ep_attention_from_scores(const float* S, const float* ML, int B, int Q, int N, float* A, ep_stream_t stream) {
  EP_REQUIRE(S && ML && A && B > 0 && Q > 0 && N > 0, EP_E_ARG, "ep_attention_from_scores: bad argument");
  return attention_from_scores(S, ML, B * Q, N, A, (hipStream_t)stream);
}

int ep_project_forward(const float* P, const float* Wv, int B, int D, int Dp, int Q, float* y, ep_stream_t stream) {
  EP_REQUIRE(P && Wv && y && B > 0 && D > 0 && Dp > 0 && Q > 0, EP_E_ARG, "ep_project_forward: bad argument");
  EP_REQUIRE(Dp % Q == 0, EP_E_SHAPE, "Dp %% Q != 0");
  return project_forward(P, Wv, B, D, Dp, Q, y, (hipStream_t)stream);
}

int ep_project_backward(const float* dy, const float* y, const float* P, const float* Wv, int B, int D, int Dp,
                        int Q, float* dP, float* dWv, float* ML, int accumulate, ep_stream_t stream) {
  EP_REQUIRE(dy && P && Wv && B > 0 && D > 0 && Dp > 0 && Q > 0, EP_E_ARG, "ep_project_backward: bad argument");
  EP_REQUIRE(Dp % Q == 0, EP_E_SHAPE, "Dp %% Q != 0");
  return project_backward(dy, y, P, Wv, B, D, Dp, Q, dP, dWv, ML, accumulate, (hipStream_t)stream);
}

size_t ep_bn_workspace_bytes(int B, int Dp) { return bn_workspace_bytes(B, Dp); }
int ep_bn_forward_train(const float* y, int B, int Dp, float eps, float momentum, float* z, float* rstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, void* workspace,
                        size_t workspace_bytes, ep_stream_t stream) {
  EP_REQUIRE(y && z && rstd && running_mean && running_var && workspace && B > 0 && Dp > 0, EP_E_ARG, "ep_bn_forward_train: bad argument");
  EP_REQUIRE(workspace_bytes >= bn_workspace_bytes(B, Dp), EP_E_WORKSPACE, "ep_bn_forward_train: workspace too small");
  return bn_forward_train(y, B, Dp, eps, momentum, z, rstd, running_mean, running_var, num_batches_tracked,
                          static_cast<float*>(workspace), (hipStream_t)stream);
}
int ep_bn_forward_eval(const float* y, int B, int Dp, float eps, const float* running_mean, const float* running_var,
                       float* z, ep_stream_t stream) {
  EP_REQUIRE(y && z && running_mean && running_var && B > 0 && Dp > 0, EP_E_ARG, "ep_bn_forward_eval: bad argument");
  return bn_forward_eval(y, B, Dp, eps, running_mean, running_var, z, (hipStream_t)stream);
}
int ep_bn_backward(const float* dz, const float* z, const float* rstd, int B, int Dp, float* dy, void* workspace,
                   size_t workspace_bytes, ep_stream_t stream) {
  EP_REQUIRE(dz && z && rstd && dy && workspace && B > 0 && Dp > 0, EP_E_ARG, "ep_bn_backward: bad argument");
  EP_REQUIRE(workspace_bytes >= bn_workspace_bytes(B, Dp), EP_E_WORKSPACE, "ep_bn_backward: workspace too small");
  return bn_backward(dz, z, rstd, B, Dp, dy, static_cast<float*>(workspace), (hipStream_t)stream);
}

int ep_linear_forward(const float* z, const float* Wc, const float* bc, int B, int Dp, int C, float* logits, int ldl,
                      ep_stream_t stream) {
  EP_REQUIRE(z && Wc && logits && B > 0 && Dp > 0 && C > 0 && ldl >= C, EP_E_ARG, "ep_linear_forward: bad argument");
  return linear_forward(z, Wc, bc, B, Dp, C, logits, ldl, (hipStream_t)stream);
}
int ep_linear_backward(const float* dlogits, int ldl, const float* z, const float* Wc, int B, int Dp, int C,
                       float* dz, float* dWc, float* dbc, int accumulate, ep_stream_t stream) {
  EP_REQUIRE(dlogits && z && Wc && B > 0 && Dp > 0 && C > 0 && ldl >= C, EP_E_ARG, "ep_linear_backward: bad argument");
  return linear_backward(dlogits, ldl, z, Wc, B, Dp, C, dz, dWc, dbc, accumulate, (hipStream_t)stream);
}

size_t ep_planes_elems(int rows, int K) { return rows > 0 && K > 0 ? planes_elems(rows, K) : 0; }
int ep_planes_split(const float* W, int R, int K, int64_t ldw, uint16_t* planes_n, uint16_t* planes_t, ep_stream_t stream) {
  EP_REQUIRE(W && R > 0 && K > 0 && ldw >= K && (planes_n || planes_t), EP_E_ARG, "ep_planes_split: bad argument");
  PlaneSpec sp{W, R, K, ldw, planes_n, planes_t};
  return planes_split(&sp, 1, (hipStream_t)stream);
}
int ep_matmul_planes(const float* A, int64_t lda, const uint16_t* planes, int rows_w, int K, const float* bias, int M,
                     int N, float* C, int64_t ldc, ep_stream_t stream) {
  EP_REQUIRE(A && planes && C && M > 0 && N > 0 && K > 0 && N <= rows_w && lda >= K && ldc >= N, EP_E_ARG,
             "ep_matmul_planes: bad argument");
  GemmParams g{};
  g.A = A; g.lda = lda; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.bias = bias;
  g.Bpl = planes; g.ldbp = (int64_t)round_up((size_t)K, 32); g.pl_term = (int64_t)rows_w * g.ldbp;
  return gemm_planes(g, 1, (hipStream_t)stream);
}

int ep_cross_entropy(const float* logits, int ldl, const int64_t* targets, int B, int C, float grad_scale,
                     float* row_stats, float* dlogits, float* stats, ep_stream_t stream) {
  EP_REQUIRE(logits && targets && B > 0 && C > 0 && ldl >= C, EP_E_ARG, "ep_cross_entropy: bad argument");
  EP_REQUIRE(row_stats || !stats, EP_E_ARG, "ep_cross_entropy: stats needs the row_stats scratch (B*4 floats)");
  EP_REQUIRE(!row_stats || aligned16(row_stats), EP_E_ALIGN, "row_stats must be 16-byte aligned");
  EP_TRY(cross_entropy(logits, ldl, targets, B, C, grad_scale, nullptr, dlogits, row_stats, (hipStream_t)stream));
  if (stats) EP_TRY(ce_stats(row_stats, B, stats, (hipStream_t)stream));
  return 0;
}

size_t ep_optim_workspace_bytes(int64_t total_numel, int num_segments) {
  return optim_workspace_bytes(total_numel, num_segments);
}
int ep_lars_step(float* params, const float* grads, float* mu, int64_t total_numel, const ep_segment* segs_host,
                 int num_segments, float lr, float weight_decay, float momentum, float trust_coefficient,
                 float inv_scale, int32_t* found_inf, float* grad_norm_out, void* workspace, size_t workspace_bytes,
                 ep_stream_t stream) {
  return optim_step(0, params, grads, mu, nullptr, total_numel, segs_host, num_segments, lr, weight_decay, momentum,
                    trust_coefficient, inv_scale, 0.f, 0.f, 0.f, 0, found_inf, grad_norm_out, workspace,
                    workspace_bytes, (hipStream_t)stream);
}
int ep_sgd_step(float* params, const float* grads, int64_t total_numel, float lr, float weight_decay, float inv_scale,
                int32_t* found_inf, float* grad_norm_out, void* workspace, size_t workspace_bytes, ep_stream_t stream) {
  return optim_step(1, params, grads, nullptr, nullptr, total_numel, nullptr, 0, lr, weight_decay, 0.f, 0.f, inv_scale,
                    0.f, 0.f, 0.f, 0, found_inf, grad_norm_out, workspace, workspace_bytes, (hipStream_t)stream);
}
int ep_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t total_numel,
                  int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float inv_scale,
                  int32_t* found_inf, float* grad_norm_out, void* workspace, size_t workspace_bytes,
                  ep_stream_t stream) {
  return optim_step(2, params, grads, exp_avg, exp_avg_sq, total_numel, nullptr, 0, lr, weight_decay, 0.f, 0.f,
                    inv_scale, beta1, beta2, eps, step, found_inf, grad_norm_out, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

int64_t ep_head_param_offsets(const ep_head_dims* d, int64_t offsets[4]) {
  const int64_t Dp = d->D / d->d_out;
  const int64_t sizes[4] = {(int64_t)d->Q * d->D, Dp * d->D, (int64_t)d->C * Dp, (int64_t)d->C};
  int64_t off = 0;
  for (int i = 0; i < 4; ++i) {
    offsets[i] = off;
    off += (sizes[i] + 3) / 4 * 4;
  }
  return off;
}

size_t ep_head_workspace_bytes(const ep_head_dims* dims) {
  if (!dims || check_dims(*dims) != 0) return 0;
  return carve(*dims, nullptr).total;
}

int64_t ep_head_workspace_flag_offset(const ep_head_dims* dims) {
  if (!dims || check_dims(*dims) != 0) return -1;
  char* base = reinterpret_cast<char*>(uintptr_t(1) << 20);   // carve() only does address arithmetic on a non-null base
  const HeadWs w = carve(*dims, base);
  return reinterpret_cast<char*>(w.iperr) - base;
}

int64_t ep_head_workspace_logits_offset(const ep_head_dims* dims, int32_t* ldl) {
  if (!dims || check_dims(*dims) != 0) return -1;
  char* base = reinterpret_cast<char*>(uintptr_t(1) << 20);
  const HeadWs w = carve(*dims, base);
  if (ldl) *ldl = w.ldl;
  return reinterpret_cast<char*>(w.logits) - base;
}

int ep_head_workspace_init(const ep_head_dims* dims, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && ws, EP_E_ARG, "ep_head_workspace_init: null pointer");
  EP_TRY(check_dims(*dims));
  const HeadWs w = carve(*dims, ws);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_head_workspace_init: workspace %zu < %zu", ws_bytes, w.total);
  // ycnt | dcnt | ticket | give-up count are one contiguous block (carve)
  EP_HIP(hipMemsetAsync(w.ycnt, 0, (2 * (size_t)w.nrb * 32 + 64) * sizeof(int), (hipStream_t)stream));
  return 0;
}

int ep_head_train_step(const ep_head_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_head_train_step: null pointer");
  const ep_head_dims& d = s->dims;
  EP_TRY(check_dims(d));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  EP_REQUIRE(s->arith == EP_ARITH_F32 || s->arith == EP_ARITH_BF16_AUTOCAST, EP_E_ARG, "ep_head_train_step: arith %d is neither EP_ARITH_F32 nor EP_ARITH_BF16_AUTOCAST", s->arith);
  // the arithmetic mode of every contraction this call enqueues (restored on every return path) -- set BEFORE the workspace
  // is carved: which plane buffers exist depends on it when EP_GEMM_PLANES=0
  const ArithScope arith_scope(s->arith);
  HeadWs w = carve(d, ws);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_head_train_step: workspace %zu < %zu%s", ws_bytes, w.total,
             s->arith == EP_ARITH_BF16_AUTOCAST ? " (the AMP-bf16 mode needs the weight planes: size the workspace with EP_GEMM_PLANES unset)" : "");
  if (s->arith == EP_ARITH_BF16_AUTOCAST) {
    EP_REQUIRE(head_planes_mode(d) == 1 && w.plWv && w.plWc, EP_E_UNSUPPORTED,
               "ep_head_train_step: the AMP-bf16 arithmetic mode runs its contractions against the weight planes (D and D / d_out multiples of 4; D=%d d_out=%d Q=%d)", d.D, d.d_out, d.Q);
    // (round 6: the split phases 4 / 8 of the pipelined data-parallel schedule run the mode too -- every phase call sets the
    // arithmetic for the contractions IT enqueues; tests/test_gpu_overlap.py.  The deferred large update stays fp32-only: untested.)
    EP_REQUIRE((s->phases & (16 | 32)) == 0, EP_E_UNSUPPORTED, "ep_head_train_step: the AMP-bf16 arithmetic mode does not take the deferred large update (phases 16 / 32)");
  }
  ScalerDev scd{};
  const ScalerDev* scaler = nullptr;
  if (s->scaler_state) {
    EP_REQUIRE(s->scaler_slot == 0 || s->scaler_slot == 1, EP_E_ARG, "ep_head_train_step: scaler_slot %d", s->scaler_slot);
    EP_REQUIRE(s->opt_num_segments == 0 && (s->phases & (16 | 32)) == 0, EP_E_UNSUPPORTED,
               "ep_head_train_step: the device-resident loss scale needs whole-tensor, undeferred optimizer phases");
    EP_REQUIRE(s->scaler_interval > 0 && s->scaler_growth > 0.f && s->scaler_backoff > 0.f, EP_E_ARG, "ep_head_train_step: scaler growth / backoff / interval must be positive");
    scd = ScalerDev{s->scaler_state, s->scaler_slot, s->scaler_growth, s->scaler_backoff, s->scaler_interval};
    scaler = &scd;
  }
  const float* scale_dev = scaler ? s->scaler_state + 2 * s->scaler_slot : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int Dp = d.D / d.d_out;
  int64_t offs[4];
  const int64_t total = ep_head_param_offsets(&d, offs);
  float* cls = s->params + offs[0]; float* Wv = s->params + offs[1];
  float* Wc = s->params + offs[2]; float* bc = s->params + offs[3];
  const float scale = (float)pow((double)d.D, -0.5);   // head_dim ** -0.5 with num_heads = 1 (ep.py:19-20)
  if (s->phases & (1 | 4 | 8)) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, d.Q));
  }
  PoolParams p = pool_params(s->x, s->x_bstride, d.B, d.N, d.D, d.Q, scale, s->x_dtype);
  p.cls = cls; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = s->image_index;
  p.nterms = s->arith == EP_ARITH_BF16_AUTOCAST ? 1 : 0;   // (bf16-token matrix-core passes: single product; the others ignore it)
  // Weight planes of THIS step (ep_planes.hip).  (In front of the first pass on the same stream instead: 2.377 against 2.385 ms
  // at 196 x 4096 -- the same.)  With the whole step in one call and an aux stream the split runs BESIDE
  // the first token pass (it needs the weights only, the pass the queries only): fork before the pass is enqueued, join
  // in front of the first contraction.  Split phases (data-parallel overlap): the large update of the previous step
  // lands between phase 4 and phase 8, so the split runs in phase 8 on the main stream.
  const bool plc = head_planes_ok(d) && (s->phases & (1 | 8));           // classifier contractions on the planes kernel
  const bool pl = plc && head_planes_mode(d) == 1;                        // ... and the two projections
  const bool pl_dp = pl && head_planes_dp(d);                             // dP against the planes of Wv^T (slices of whole k-groups only)
  hipEvent_t pev[2] = {nullptr, nullptr};
  bool split_done = false;
  if (plc && s->planes_valid) {
    split_done = true;                                 // written by the previous optimizer phase (ep_optim.hip: tile_update_emit)
  } else if (plc && (s->phases & 1) && s->aux_stream && (hipStream_t)s->aux_stream != st) {
    hipStream_t ax = (hipStream_t)s->aux_stream;
    hipEvent_t evs[6];
    EP_TRY(get_events(evs, 6));
    pev[0] = evs[4]; pev[1] = evs[5];               // (events 0..2 are the fork / join of the weight-gradient side path)
    EP_HIP(hipEventRecord(pev[0], st));              // after the previous optimizer update
    EP_HIP(hipStreamWaitEvent(ax, pev[0], 0));
    EP_TRY(head_planes_split(d, w, Wv, Wc, ax));
    EP_HIP(hipEventRecord(pev[1], ax));
    split_done = true;
  }
  DeferredReduce red{};                                       // last stage of the dcls reduction, finished by the optimizer
  // In-pass contractions (ep_inpass.h): y inside the first pass -- only when the whole forward/backward is this call: in
  // the split schedule (phases 4 / 8) the large update of the previous step lands BETWEEN the first pass and the
  // projection, so the pass must not read Wv -- and dP inside the second pass.
  const int ipmask = (s->phases & (1 | 4 | 8)) && !pl ? pool_inpass_mask(p, Dp) : 0;
  const bool ip_y = (ipmask & 1) && (s->phases & 1) && bn_takes_parts(d.B);
  int ip_r0 = 0;
  if (ipmask) {
    p.ip_err = w.iperr;
    p.ip_zero = w.dcnt; p.ip_nzero = w.nrb * 32 + 32;                   // the first pass clears the second pass's counters
    if (ip_y) {
      p.ip_WvF = Wv; p.ip_ypart = w.ypart; p.ip_ycnt = w.ycnt; p.ip_y = w.y;
      // rows of images that end a round before the pass does get their y whole (hidden under the last round)
      const StreamGridInfo gi = pool_stream_grid(p);
      // EP_INPASS_YSPLIT=1 only: measured WORSE than K-quarter tasks for every row block (first pass 171 against 159 us at
      // 256x768 -- a whole-y task takes ~28 us and the helpers only get to it ~25 us before the pass ends)
      static int ysplit = -1;
      if (ysplit < 0) { const char* e = getenv("EP_INPASS_YSPLIT"); ysplit = e ? atoi(e) : 0; }
      ip_r0 = ysplit && gi.rounds > 1 && gi.helpers > 0 ? bn_parts_r0(d.B, (gi.rounds - 1) * gi.grid) : 0;
      p.ip_yr0 = ip_r0;
    }
  }
  // a deferred large update of the previous step (phases bit 5): v.weight / fc.* are first read behind the first token
  // pass -- unless that pass computes the projection itself
  const bool wait_defer = (s->phases & 32) && s->defer_event;
  if (wait_defer && ip_y) EP_HIP(hipStreamWaitEvent(st, (hipEvent_t)s->defer_event, 0));
  if (s->phases & (1 | 4)) {                                  // first token pass: depends on cls_token only
    mark_pass(0, st);
    EP_TRY(pool_forward(p, st));
    mark_pass(1, st);
  }
  if (wait_defer && !ip_y) EP_HIP(hipStreamWaitEvent(st, (hipEvent_t)s->defer_event, 0));
  const bool plw = pl && (s->phases & 1) && head_planes_wgrad(d) && w.ptP;     // weight gradients on the planes kernel
  p.ip_WvF = nullptr; p.ip_ypart = nullptr; p.ip_ycnt = nullptr; p.ip_y = nullptr; p.ip_yr0 = 0; p.ip_zero = nullptr; p.ip_nzero = 0;
  if (s->phases & (1 | 8)) {
    if (plc) {
      if (split_done) { if (pev[1]) EP_HIP(hipStreamWaitEvent(st, pev[1], 0)); }
      else EP_TRY(head_planes_split(d, w, Wv, Wc, st));
    }
    if (pl) {
      // thin query slices (<= 32 columns: the protocol's 32 queries): the 64 x 32-tile kernel of ep_gemm.hip -- in the AMP-bf16 mode,
      // the one that reaches here with such slices, as a single product -- instead of 64-column planes tiles that are half padding
      if (Dp / d.Q <= 32 && d.B >= 64) EP_TRY(project_forward(w.P, Wv, d.B, d.D, Dp, d.Q, w.y, st));
      else EP_TRY(project_forward_pl(w, d, st));
      // the planes of P^T (for dWv) are split on the side stream behind the value projection -- beside BatchNorm, logits, CE,
      // dz: started right behind the first pass the HBM-bound split slows the projection, which reads P too, from 225 to 331 us
      if (plw) {
        hipStream_t side = s->aux_stream ? (hipStream_t)s->aux_stream : st;
        if (side != st) {
          hipEvent_t ev4[4];
          EP_TRY(get_events(ev4, 4));
          EP_HIP(hipEventRecord(ev4[3], st));
          EP_HIP(hipStreamWaitEvent(side, ev4[3], 0));
        }
        EP_TRY(wgrad_split_P(w, d, side));
      }
    } else if (!ip_y) {
      EP_TRY(project_forward(w.P, Wv, d.B, d.D, Dp, d.Q, w.y, st));
    }
    if (ip_y)   // the K-quarter partials of the in-pass projection are summed by the BatchNorm kernel (which also writes y)
      EP_TRY(bn_forward_train(w.ypart, d.B, Dp, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                              s->num_batches_tracked, w.bnpart, st, IP_YPARTS, (int64_t)d.B * Dp, w.y, ip_r0));
    else
    EP_TRY(bn_forward_train(w.y, d.B, Dp, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    if (plc) EP_TRY(linear_forward_pl(w, d, bc, st));
    else EP_TRY(linear_forward(w.z, Wc, bc, d.B, Dp, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st, scale_dev));
    // The weight gradients dWc / dbc / dWv and the statistics feed nothing before the optimizer.
    // Preferred: they ride in the launch of the second token pass as extra workgroups, which the
    // dispatcher places as pooling workgroups retire (the tail of the pass) -- no second stream, no
    // cross-queue events, no contention with the critical-path contractions.  Otherwise (kernel
    // families without side support, unaligned shapes) they run on the aux stream beside it.
    const GemmParams gWc = dwc_gemm(w.dlogits, w.ldl, w.z, d.B, Dp, d.C, s->grads + offs[2], s->accumulate);
    const GemmParams gWv = dwv_gemm(w.dy, w.P, d.B, d.D, Dp, d.Q, s->grads + offs[1], s->accumulate);
    p.dP = w.dP; p.Gpart = static_cast<float*>(w.pool_ws);
    if (ipmask) { p.ip_zero = w.ycnt; p.ip_nzero = w.nrb * 32; }            // the second pass clears the first pass's counters
    // Round 6, the default wherever the classifier's contractions run on the planes kernel: the side work leaves the second token
    // pass.  dWc = dlogits^T z, the bias column sum and the statistics fold ride in the launch of dz = dlogits Wc (their
    // operands exist behind the loss kernel), dWv_q = dy_q^T P_q in the launch of dP = dy_q Wv_q (mode 1; otherwise it stays
    // with the pass) -- ep_planes.hip: ep_gemm_planes_side_kernel.  The pass then streams alone.  EP_CHAIN=0: the round-5 schedule.
    // EP_WG2=1 (round 6 experiment): the two weight gradients as ONE launch of the paired-group tile between BatchNorm backward and
    // the second pass (ep_gemm.hip: wgrad_pair) instead of side workgroups in the pass, which keeps only the bias sum and the
    // statistics fold
    static int wg2_on = -1;
    if (wg2_on < 0) { const char* e = getenv("EP_WG2"); wg2_on = e ? atoi(e) : 0; }
    const bool wg2 = wg2_on && wgrad_pair_ok(gWc) && wgrad_pair_ok(gWv);
    static int chain_on = -1;
    if (chain_on < 0) { const char* e = getenv("EP_CHAIN"); chain_on = e ? atoi(e) : 0; }
    if (chain_on && plc && gemm_side_ok(gWc, false, false) && gemm_side_ok(gWv, false, false)) {
      SideTasks sd1{};
      side_add_gemm(sd1, gWc, 1);
      sd1.cs_src = w.dlogits; sd1.cs_out = s->grads + offs[3]; sd1.cs_B = d.B; sd1.cs_ncol = d.C; sd1.cs_ld = w.ldl;
      sd1.cs_accumulate = s->accumulate; sd1.n_colsum = (d.C + 15) / 16;
      sd1.rowstat = w.rowstat; sd1.stats = s->stats; sd1.rs_B = d.B; sd1.n_stats = 1;
      sd1.total += sd1.n_colsum + sd1.n_stats;
      EP_TRY(gemm_planes_side(planes_gemm(w.dlogits, w.ldl, 0, w.plWcT, Dp, d.C, 0, w.dz, Dp, 0, d.B, Dp, d.C, nullptr), 1, sd1, st));
      EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, Dp, w.dy, w.bnpart, st));
      const bool in_pass = pool_backward_takes_delta(p, Dp);
      const bool ip_dp = (ipmask & 2) != 0;                              // (mode 2 only: the mask is 0 when the projections run on the planes)
      if (in_pass) { p.dyv = w.dy; p.yv = w.y; p.Dv = Dp; }
      if (ip_dp) { p.ip_dy = w.dy; p.ip_Wv = Wv; p.ip_dcnt = w.dcnt; }
      if ((ipmask & 4) && in_pass) p.tick = w.tick;
      SideTasks sd2{};
      side_add_gemm(sd2, gWv, d.Q);
      const bool pass_side = pool_backward_takes_side(p);
      hipStream_t side = s->aux_stream ? (hipStream_t)s->aux_stream : st;
      hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
      if (pl_dp) {
        if (!in_pass) EP_TRY(delta_rows(w.dy, w.y, d.B * d.Q, Dp / d.Q, w.ML, st));
        const int Dq = Dp / d.Q;
        EP_TRY(gemm_planes_side(planes_gemm(w.dy, Dp, Dq, w.plWvT, d.D, Dp, Dq, w.dP, (int64_t)d.Q * d.D, d.D, d.B, d.D, Dq, nullptr), d.Q, sd2, st));
      } else {
        if (!pass_side) {                                                // dWv on the side queue, started in front of dP
          if (side != st) {
            EP_TRY(get_events(ev, 4));
            EP_HIP(hipEventRecord(ev[1], st));
            EP_HIP(hipStreamWaitEvent(side, ev[1], 0));
          }
          EP_TRY(project_backward(w.dy, nullptr, w.P, Wv, d.B, d.D, Dp, d.Q, nullptr, s->grads + offs[1], nullptr, s->accumulate, side));
        }
        EP_TRY(project_backward(w.dy, w.y, w.P, Wv, d.B, d.D, Dp, d.Q, ip_dp ? nullptr : w.dP, nullptr, in_pass ? nullptr : w.ML, 0, st));
      }
      mark_pass(2, st);
      EP_TRY(pool_backward(p, s->grads + offs[0], s->accumulate, st, (!pl_dp && pass_side) ? &sd2 : nullptr, (s->phases & 2) ? &red : nullptr));
      mark_pass(3, st);
      if (!pl_dp && !pass_side && side != st) {
        EP_HIP(hipEventRecord(ev[2], side));
        EP_HIP(hipStreamWaitEvent(st, ev[2], 0));                        // join: grads complete on `stream`
      }
    } else
    if (pool_backward_takes_side(p) && gemm_side_ok(gWc, false, false) && gemm_side_ok(gWv, false, false)) {
      // the softmax-correction rows dy_q . y_q: inside the second pass where its kernel can (one launch less)
      const bool in_pass = pool_backward_takes_delta(p, Dp);
      const bool ip_dp = (ipmask & 2) != 0;
      // BatchNorm backward FOLDED into the in-pass dP tasks (ep_inpass.h): the dz contraction leaves per-tile column
      // statistics in its epilogue, the tasks form dy = rstd (dz - m1 - z m2) while they stage their A tile and publish
      // dy for the delta items and the dWv side tasks -- ep_bn_bwd_fused_kernel (8 us, 48 workgroups on 256 CUs)
      // leaves the chain between the passes.  OPT-IN (EP_BN_FOLD=1): measured slower so far -- the tasks in front of the
      // token stream grow by ~9 us (one more operand tile, the tile-ordered column sums) for the 8 us launch they replace
      // (second pass 201 -> 210 us, step 0.441 -> 0.444 ms at 256x768).
      GemmParams gz{};
      gz.A = w.dlogits; gz.lda = w.ldl; gz.B = Wc; gz.ldb = Dp; gz.extB = Dp; gz.C = w.dz; gz.ldc = Dp;
      gz.M = d.B; gz.N = Dp; gz.K = d.C; gz.alpha = 1.f;
      static int fold_on = -1;
      if (fold_on < 0) { const char* e = getenv("EP_BN_FOLD"); fold_on = e ? atoi(e) : 0; }
      const bool fold = fold_on && ip_dp && in_pass && !(ipmask & 4) && !plc && d.B % 32 == 0 && s->x_dtype == EP_DTYPE_F32 && gemm_colstats_ok(true, false, gz, 1);
      if (fold) {
        gz.cs_z = w.z; gz.cs_out = w.colstat;
        EP_TRY(gemm(true, false, gz, 1, st));
        p.ip_fold_dz = w.dz; p.ip_fold_z = w.z; p.ip_fold_rstd = w.rstd; p.ip_fold_cs = w.colstat;
      } else {
        if (plc) EP_TRY(linear_backward_dz_pl(w, d, st));
        else EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, Dp, d.C, w.dz, nullptr, nullptr, 0, st));
        EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, Dp, w.dy, w.bnpart, st));
      }
      if (in_pass) { p.dyv = w.dy; p.yv = w.y; p.Dv = Dp; }
      if (ip_dp) { p.ip_dy = w.dy; p.ip_Wv = Wv; p.ip_dcnt = w.dcnt; }   // dP rows by the pooling workgroups themselves
      if ((ipmask & 4) && in_pass) p.tick = w.tick;                      // ... in the ticketed form (ep_pool_bwd2.hip)
      if (pl_dp) {
        if (!in_pass) EP_TRY(delta_rows(w.dy, w.y, d.B * d.Q, Dp / d.Q, w.ML, st));
        EP_TRY(project_backward_dP_pl(w, d, st));
      } else {
        EP_TRY(project_backward(w.dy, w.y, w.P, Wv, d.B, d.D, Dp, d.Q, ip_dp ? nullptr : w.dP, nullptr,
                                in_pass ? nullptr : w.ML, 0, st));
      }
      SideTasks sd{};
      if (wg2 && !fold) {
        const GemmParams pair[2] = {gWc, gWv};
        const int pb[2] = {1, d.Q};
        EP_TRY(wgrad_pair(pair, pb, 2, st));
      } else {
        side_add_gemm(sd, gWc, 1);
        side_add_gemm(sd, gWv, d.Q);
      }
      sd.cs_src = w.dlogits; sd.cs_out = s->grads + offs[3]; sd.cs_B = d.B; sd.cs_ncol = d.C; sd.cs_ld = w.ldl;
      sd.cs_accumulate = s->accumulate; sd.n_colsum = (d.C + 15) / 16;
      sd.rowstat = w.rowstat; sd.stats = s->stats; sd.rs_B = d.B; sd.n_stats = 1;
      sd.total += sd.n_colsum + sd.n_stats;
      mark_pass(2, st);
      EP_TRY(pool_backward(p, s->grads + offs[0], s->accumulate, st, &sd, (s->phases & 2) ? &red : nullptr));
      mark_pass(3, st);
    } else {
      hipStream_t side = s->aux_stream ? (hipStream_t)s->aux_stream : st;
      hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
      // dWv on a queue of its own where the aux stream's work would otherwise push it into the second pass (EP_SIDE2=0: one side stream)
      static int side2_on = -1;
      if (side2_on < 0) { const char* e = getenv("EP_SIDE2"); side2_on = e ? atoi(e) : 1; }
      hipStream_t side2 = side;
      if (side != st && side2_on && !plw && d.Q > 16) EP_TRY(get_side2_stream(&side2));
      // ONE fork (round 6 experiment, EP_ONE_FORK: 0 = off, the default; 1 = with two side queues only; 2 = always): every event
      // recorded on the step's own stream costs it ~6 us of idle queue (rocprofv3 timelines of the 32-query steps: three records
      // and a join = ~30 us per step), so both side queues could fork once, behind BatchNorm backward.  Measured SLOWER: the
      // classifier's weight gradient loses its head start and runs into the second pass -- 196 x 1024 Q = 32 0.787 / 0.795 ->
      // 0.837 / 0.817 ms, 256 x 768 Q = 32 bf16 0.431 / 0.453 -> 0.469, fp32 0.721 / 0.713 -> 0.755 / 0.749; Q = 8 at 196 x 1024 and
      // 256 x 1152 unchanged within noise.
      static int one_fork_env = -1;
      if (one_fork_env < 0) { const char* e = getenv("EP_ONE_FORK"); one_fork_env = e ? atoi(e) : 0; }
      const bool one_fork = side != st && !plw && (one_fork_env == 2 || (one_fork_env == 1 && side2 != side));
      if (side != st) EP_TRY(get_events(ev, 4));
      if (side != st && !one_fork) {
        EP_HIP(hipEventRecord(ev[0], st));
        EP_HIP(hipStreamWaitEvent(side, ev[0], 0));
      }
      struct SideHint { int old; explicit SideHint(int v) : old(t_wgrad_side) { t_wgrad_side = v; } ~SideHint() { t_wgrad_side = old; } };
      const SideHint hint(d.Q > 16 ? 0 : 1);                // (restored when this branch is left, error returns included)
      // ... and, as launches of their own on the bf16 x3 tile (32 queries), with K slices (ep_gemm.hip): scratch per gradient
      struct KsHint { explicit KsHint(float* c, float* v, size_t n) { t_wgks[0] = c; t_wgks[1] = v; t_wgks_floats = n; }
                      ~KsHint() { t_wgks[0] = t_wgks[1] = nullptr; t_wgks_floats = 0; } };
      const KsHint kshint(side != st ? w.wgks_c : nullptr, side != st ? w.wgks_v : nullptr, w.wgks_floats);
      auto classifier_side = [&]() -> int {                  // statistics fold, dWc, dbc: nothing before the optimizer reads them
        EP_TRY(ce_stats(w.rowstat, d.B, s->stats, side));
        if (plw) {
          EP_TRY(wgrad_dwc_pl(w, d, s->grads + offs[2], s->accumulate, side));
          EP_TRY(colsum(w.dlogits, d.B, d.C, w.ldl, s->accumulate, s->grads + offs[3], side));
        } else {
          EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, Dp, d.C, nullptr, s->grads + offs[2], s->grads + offs[3],
                                 s->accumulate, side));
        }
        return 0;
      };
      if (!one_fork) EP_TRY(classifier_side());
      if (plc) EP_TRY(linear_backward_dz_pl(w, d, st));
      else EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, Dp, d.C, w.dz, nullptr, nullptr, 0, st));
      EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, Dp, w.dy, w.bnpart, st));
      if (side != st) {
        EP_HIP(hipEventRecord(ev[1], st));
        EP_HIP(hipStreamWaitEvent(side2, ev[1], 0));
        if (one_fork && side2 != side) EP_HIP(hipStreamWaitEvent(side, ev[1], 0));
      }
      if (one_fork) EP_TRY(classifier_side());
      // (The weight gradient of v started only when dP is done, so that it runs beside the HBM-bound second pass instead of
      // beside dP: measured, the pass then takes 1070 instead of 555 us at 196 x 4096 -- the two kernels do not share CUs.)
      if (plw) EP_TRY(wgrad_dwv_pl(w, d, s->grads + offs[1], s->accumulate, side));
      else EP_TRY(project_backward(w.dy, nullptr, w.P, Wv, d.B, d.D, Dp, d.Q, nullptr, s->grads + offs[1], nullptr,
                                   s->accumulate, side2));
      if (side2 != side) EP_HIP(hipEventRecord(ev[3], side2));
      // the softmax-correction rows dy_q . y_q inside the second pass where its kernel can (the 32-query bf16 pass): one launch less
      const bool in_pass2 = pool_backward_takes_delta(p, Dp);
      if (in_pass2) { p.dyv = w.dy; p.yv = w.y; p.Dv = Dp; }
      if (pl_dp) {
        if (!in_pass2) EP_TRY(delta_rows(w.dy, w.y, d.B * d.Q, Dp / d.Q, w.ML, st));
        EP_TRY(project_backward_dP_pl(w, d, st));
      } else {
        EP_TRY(project_backward(w.dy, w.y, w.P, Wv, d.B, d.D, Dp, d.Q, w.dP, nullptr, in_pass2 ? nullptr : w.ML, 0, st));
      }
      mark_pass(2, st);
      EP_TRY(pool_backward(p, s->grads + offs[0], s->accumulate, st, nullptr, (s->phases & 2) ? &red : nullptr));
      mark_pass(3, st);
      if (side != st) {
        EP_HIP(hipEventRecord(ev[2], side));
        EP_HIP(hipStreamWaitEvent(st, ev[2], 0));           // join: grads complete on `stream`
        if (side2 != side) EP_HIP(hipStreamWaitEvent(st, ev[3], 0));
      }
    }
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    ep_segment segs[4];
    const int64_t sizes[4] = {(int64_t)d.Q * d.D, (int64_t)Dp * d.D, (int64_t)d.C * Dp, (int64_t)d.C};
    for (int i = 0; i < 4; ++i) segs[i] = ep_segment{offs[i], sizes[i], i < 3 ? 1 : 0, 0};   // bias: ndim 1
    const bool sub = s->opt_num_segments > 0;
    EP_REQUIRE(!sub || (s->opt_first_segment >= 0 && s->opt_first_segment + s->opt_num_segments <= 4), EP_E_ARG,
               "optimizer segment range [%d, +%d) outside the four tensors", s->opt_first_segment, s->opt_num_segments);
    // always per tensor (SGD / AdamW do not need the segments for their arithmetic -- no trust ratio -- but the update
    // kernel finds the weight matrices whose planes it writes among them)
    const ep_segment* use = segs + (sub ? s->opt_first_segment : 0);
    const int nuse = sub ? s->opt_num_segments : 4;
    hipStream_t ax = (hipStream_t)s->aux_stream;
    // a give-up of an in-pass hand-off wait (w.iperr != 0: ep_inpass.h) must be loud: the optimizer then skips the update,
    // sets found_inf and bumps the non-finite row count of the step statistics, which stops train_one_epoch
    float* abort_stat = s->stats ? s->stats + 3 : nullptr;
    // the planes of the matrices this phase updates are written by the update itself (no split launch in the next step)
    PlaneSpec emit[2];
    int n_emit = 0;
    if (head_planes_ok(d)) {
      emit[n_emit++] = PlaneSpec{Wc, d.C, Dp, Dp, w.plWc, w.plWcT};
      if (head_planes_mode(d) == 1) emit[n_emit++] = PlaneSpec{Wv, Dp, d.D, d.D, w.plWv, w.plWvT};
    }
    if ((s->phases & 16) && s->defer_event && ax && ax != st && !sub) {
      // deferred large update: cls_token here (one launch; it also finishes the cls_token gradient reduction), the three
      // large tensors on the aux stream beside whatever the caller enqueues next on `stream`
      EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total, segs, 1, s->lr,
                        s->weight_decay, s->momentum, s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps,
                        s->opt_step, s->found_inf, s->grad_norm, w.opt_ws, w.opt_ws_bytes, st, &red, w.iperr, abort_stat, emit, n_emit));
      hipEvent_t evs[6];
      EP_TRY(get_events(evs, 6));
      EP_HIP(hipEventRecord(evs[3], st));                      // the gradients are complete on `stream` here
      EP_HIP(hipStreamWaitEvent(ax, evs[3], 0));
      EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total, segs + 1, 3, s->lr,
                        s->weight_decay, s->momentum, s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps,
                        s->opt_step, s->found_inf, s->grad_norm, w.opt_ws, w.opt_ws_bytes, ax, nullptr, w.iperr, abort_stat, emit, n_emit));
      EP_HIP(hipEventRecord((hipEvent_t)s->defer_event, ax));
    } else
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      use, nuse, s->lr, s->weight_decay,
                      s->momentum, s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step,
                      s->found_inf, s->grad_norm, w.opt_ws, w.opt_ws_bytes, st, &red, w.iperr, abort_stat, emit, n_emit, scaler));
  }
  return 0;
}

int ep_head_eval_forward(const ep_head_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* params,
                         const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                         void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_head_eval_forward: null pointer");
  const ep_head_dims& d = *dims;
  EP_TRY(check_dims(d));
  EP_TRY(check_tokens_fwd(x, x_dtype, x_bstride, d.B, d.N, d.D, d.Q));
  HeadWs w = carve(d, ws);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  const int Dp = d.D / d.d_out;
  int64_t offs[4];
  ep_head_param_offsets(&d, offs);
  const float scale = (float)pow((double)d.D, -0.5);
  PoolParams p = pool_params(x, x_bstride, d.B, d.N, d.D, d.Q, scale, x_dtype);
  p.cls = params + offs[0]; p.cls_bstride = 0; p.P = w.P; p.S = w.S; p.ML = w.ML; p.index = image_index;
  EP_TRY(pool_forward(p, st));
  // the value projection writes z itself: eval-mode BatchNorm is a fixed per-column map, folded into the contraction's epilogue
  // (ep_side.h: store_acc_blocks) -- the arithmetic of ep_bn_eval_kernel, one launch less (EP_EVAL_FOLD=0: the round-5 chain)
  static int fold = -1;
  if (fold < 0) { const char* e = getenv("EP_EVAL_FOLD"); fold = e ? atoi(e) : 1; }
  if (fold) {
    const int Dq = Dp / d.Q;
    GemmParams g{};
    g.A = w.P; g.lda = (int64_t)d.Q * d.D; g.sAz = d.D;
    g.B = params + offs[1]; g.ldb = d.D; g.sBz = (int64_t)Dq * d.D;
    g.C = w.z; g.ldc = Dp; g.sCz = Dq;
    g.M = d.B; g.N = Dq; g.K = d.D; g.alpha = 1.f; g.extA = d.D; g.extB = d.D;
    g.bn_rm = running_mean; g.bn_rv = running_var; g.bn_eps = bn_eps; g.sBiasz = Dq;
    EP_TRY(gemm(true, true, g, d.Q, st));
  } else {
    EP_TRY(project_forward(w.P, params + offs[1], d.B, d.D, Dp, d.Q, w.y, st));
    EP_TRY(bn_forward_eval(w.y, d.B, Dp, bn_eps, running_mean, running_var, w.z, st));
  }
  EP_TRY(linear_forward(w.z, params + offs[2], params + offs[3], d.B, Dp, d.C, logits, ldl, st));
  return 0;
}


/* ---- plain linear probing: BatchNorm1d + Linear on (B, D) features ---------------------------------- */
struct LpWs { float *z, *rstd, *logits, *dlogits, *rowstat, *bnpart; void* opt_ws; size_t opt_ws_bytes; int ldl; size_t total; };

static LpWs lp_carve(const ep_head_dims& d, void* base) {
  LpWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  w.ldl = (d.C + 3) / 4 * 4;
  const size_t B = d.B;
  w.z = take(B * d.D); w.rstd = take(d.D); w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl);
  w.rowstat = take(B * 4); w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
  int64_t offs[2];
  w.opt_ws_bytes = optim_workspace_bytes(ep_lp_param_offsets(&d, offs), 2);
  w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  w.total = off;
  return w;
}

int64_t ep_lp_param_offsets(const ep_head_dims* d, int64_t offsets[2]) {
  offsets[0] = 0;
  offsets[1] = ((int64_t)d->C * d->D + 3) / 4 * 4;
  return offsets[1] + ((int64_t)d->C + 3) / 4 * 4;
}

size_t ep_lp_workspace_bytes(const ep_head_dims* dims) {
  if (!dims || dims->B <= 0 || dims->D <= 0 || dims->C <= 0 || dims->D % 4 != 0) {
    set_error("linear probe dims: B, D, C must be positive and D a multiple of 4");
    return 0;
  }
  return lp_carve(*dims, nullptr).total;
}

int ep_lp_train_step(const ep_head_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_lp_train_step: null pointer");
  const ep_head_dims& d = s->dims;
  EP_REQUIRE(d.B > 0 && d.D > 0 && d.C > 0 && d.D % 4 == 0, EP_E_SHAPE, "linear probe dims: B, D, C must be positive and D a multiple of 4");
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const LpWs w = lp_carve(d, ws);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_lp_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  EP_REQUIRE(s->arith == EP_ARITH_F32 || s->arith == EP_ARITH_BF16_AUTOCAST, EP_E_ARG, "ep_head_train_step: arith %d is neither EP_ARITH_F32 nor EP_ARITH_BF16_AUTOCAST", s->arith);
  // the arithmetic mode of every contraction this call enqueues (restored on every return path)
  const ArithScope arith_scope(s->arith);
  // (BatchNorm + Linear on pooled features: the plain linear probe has no planes and runs fp32 only)
  EP_REQUIRE(s->arith == EP_ARITH_F32, EP_E_UNSUPPORTED, "ep_lp_train_step: the AMP-bf16 arithmetic mode is implemented for the EP head's fused step only");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[2];
  const int64_t total = ep_lp_param_offsets(&d, offs);
  float* Wc = s->params + offs[0]; float* bc = s->params + offs[1];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats, EP_E_ARG, "train step: null input");
    EP_REQUIRE(s->x_dtype == EP_DTYPE_F32 && s->x_bstride == d.D && aligned16(s->x), EP_E_SHAPE,
               "linear probe: features must be a dense 16-byte aligned (B, D) fp32 matrix");
    const float* x = static_cast<const float*>(s->x);
    EP_TRY(bn_forward_train(x, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, nullptr, s->grads + offs[0], s->grads + offs[1],
                           s->accumulate, st));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    ep_segment segs[2] = {{offs[0], (int64_t)d.C * d.D, 1, 0}, {offs[1], (int64_t)d.C, 0, 0}};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? 2 : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_lp_eval_forward(const ep_head_dims* dims, const float* x, const float* params, const float* running_mean,
                       const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                       ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && logits && ws, EP_E_ARG, "ep_lp_eval_forward: null pointer");
  const ep_head_dims& d = *dims;
  EP_REQUIRE(d.B > 0 && d.D > 0 && d.C > 0 && d.D % 4 == 0 && ldl >= d.C, EP_E_SHAPE, "ep_lp_eval_forward: bad dims");
  const LpWs w = lp_carve(d, ws);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_lp_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[2];
  ep_lp_param_offsets(&d, offs);
  EP_TRY(bn_forward_eval(x, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[0], params + offs[1], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
