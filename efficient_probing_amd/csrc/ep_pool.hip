// EP attentive pooling: the two streaming passes over the frozen tokens (gfx950 / CDNA4).
//
//   forward  (reference poolings/ep.py:35-44):  S = (cls*scale) x^T ; A = softmax_n S ; P = A x
//   backward (autograd of the same lines)     :  dA = dP x^T ; dS = A (dA - delta) ;
//                                                dcls = scale * sum_b dS x
//
// Design (DESIGN.md section "pool kernels"):
//   * one workgroup streams whole images; the image's tokens are copied HBM -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of NSLOT tiles of TT
//     tokens, several tiles ahead of the compute, so x is read from HBM exactly once per pass
//     and never touches a VGPR on the way in;
//   * the Q queries are split over the NW waves of the workgroup (QW queries per wave); every
//     wave reads every token row from LDS (lane = 16-byte chunk of the row, conflict-free
//     ds_read_b128) and owns the FULL D-dimension for its queries, so there is no cross-wave
//     reduction at all -- the only synchronisation is one s_barrier per tile for the ring;
//   * scores: lane-local FMAs, then a v_permlane32_swap / v_permlane16_swap / DPP butterfly that
//     leaves score (q, t) replicated in the 16 lanes of row t of register q;
//   * softmax: lazy-max online softmax evaluated lane-parallel (row = token), weights broadcast
//     to SGPRs with v_readlane and used as the scalar operand of the pooling FMAs.
//   * LDS-DMA completion is tracked with counted s_waitcnt vmcnt(N) (never 0 in steady state).
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"
#include "ep_pool_imgq.h"

namespace ep {

constexpr float LOG2E = 1.4426950408889634f;

// out[g*n + j] = (accumulate ? out[j] : 0) + alpha * sum_{i in group g} part[i*n + j]
// grid (ceil(n/4/64), ngroups); 256 threads = 64 float4 columns x 4 partial-lanes; fixed order.
__global__ __launch_bounds__(256) void ep_reduce_partials_kernel(const float* __restrict__ part, int nparts, int n,
                                                               float alpha, int accumulate, float* __restrict__ out) {
  __shared__ f4 sm[4][64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int j = (blockIdx.x * 64 + cx) * 4;
  const int ngroups = gridDim.y, g = blockIdx.y;
  const int per = (nparts + ngroups - 1) / ngroups;
  const int i0 = g * per, i1 = (i0 + per) < nparts ? (i0 + per) : nparts;
  f4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
  if (j < n) {
    int i = i0 + py;
    // eight loads in flight per trip (a two-load trip is one memory round trip per pair); the order of the additions
    // into s0 / s1 is the one of the two-load loop below, which finishes the rest
    for (; i + 28 < i1; i += 32) {
      f4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f4*>(part + (int64_t)(i + 4 * u) * n + j);
#pragma unroll
      for (int u = 0; u < 8; u += 2) { s0 += a[u]; s1 += a[u + 1]; }
    }
    for (; i + 4 < i1; i += 8) {
      s0 += *reinterpret_cast<const f4*>(part + (int64_t)i * n + j);
      s1 += *reinterpret_cast<const f4*>(part + (int64_t)(i + 4) * n + j);
    }
    for (; i < i1; i += 4) s0 += *reinterpret_cast<const f4*>(part + (int64_t)i * n + j);
  }
  sm[py][cx] = s0 + s1;
  __syncthreads();
  if (py == 0 && j < n) {
    const f4 t = (sm[0][cx] + sm[1][cx]) + (sm[2][cx] + sm[3][cx]);
    float* o = out + (int64_t)g * n + j;
    f4 s;
    if (accumulate) {                       // ONE rounding, spelled out: ep_opt_norms_kernel's deferred stage does the same
      const f4 old = *reinterpret_cast<const f4*>(o);
      s = f4{fmaf(t.x, alpha, old.x), fmaf(t.y, alpha, old.y), fmaf(t.z, alpha, old.z), fmaf(t.w, alpha, old.w)};
    } else {
      s = t * alpha;
    }
    *reinterpret_cast<f4*>(o) = s;
  }
}

// ---------------------------------------------------------------------------------------
// generic fallback (any Q, any D % 4 == 0): one workgroup per image, two passes over the image
// (the second one is served by L2 / Infinity Cache).  Used for shapes the streaming kernel
// does not cover (D % 64 != 0, D > 1536, Q > 32) and as an in-library cross-check.
// ---------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void ep_pool_fwd_generic_kernel(PoolParams p) {
  extern __shared__ float sm[];       // scores of one query row: N floats, then 8 scratch
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int f16 = BF16 && p.x_bf16 == 2;     // fp16-stored tokens (forward only)
  const char* xb = reinterpret_cast<const char*>(p.x) + EP_IMG_OFF(p, b) * (BF16 ? 2 : 4);
  float* red = sm + N;
  // LayerNorm-of-tokens mode: per-token {mean, rstd}; scores q.xhat = rstd (q.x - mean sum(q))
  const float* ts = p.tokstat ? p.tokstat + (int64_t)(p.index ? p.index[b] : b) * N * 2 : nullptr;
  for (int q = 0; q < Q; ++q) {
    const float* cq = p.cls + (int64_t)b * p.cls_bstride + (int64_t)q * D;
    float wsum = 0.f;
    if (ts) {
      float t = 0.f;
      for (int d = threadIdx.x; d < D; d += 256) t += cq[d] * p.scale;
      t = wave_sum(t);
      if (lane == 0) red[w] = t;
      __syncthreads();
      wsum = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    }
    for (int n = w; n < N; n += 4) {
      float s = 0.f;
      for (int d = lane * 4; d < D; d += 256) {
        f4 xv = load_tok4<BF16>(xb, (int64_t)n * D + d, f16);
        f4 cv = *reinterpret_cast<const f4*>(cq + d) * p.scale;
        s = fmaf(cv.x, xv.x, s); s = fmaf(cv.y, xv.y, s); s = fmaf(cv.z, xv.z, s); s = fmaf(cv.w, xv.w, s);
      }
      s = wave_sum(s);
      if (ts) s = ts[2 * n + 1] * (s - ts[2 * n] * wsum);
      if (p.sbias) s += p.sbias[((int64_t)b * Q + q) * N + n];
      if (lane == 0) { sm[n] = s; p.S[((int64_t)b * Q + q) * N + n] = s; }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int n = threadIdx.x; n < N; n += 256) mx = fmaxf(mx, sm[n]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float l = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
      float e = __builtin_amdgcn_exp2f((sm[n] - mx) * LOG2E);
      sm[n] = e; l += e;
    }
    l = wave_sum(l);
    if (lane == 0) red[w] = l;
    __syncthreads();
    l = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / l;
    float shift = 0.f;                                 // sum_n e_n rstd_n mean_n (LayerNorm mode)
    if (ts) {
      __syncthreads();
      float c = 0.f;
      for (int n = threadIdx.x; n < N; n += 256) { const float er = sm[n] * ts[2 * n + 1]; c = fmaf(er, ts[2 * n], c); sm[n] = er; }
      c = wave_sum(c);
      if (lane == 0) red[w] = c;
      __syncthreads();
      shift = (red[0] + red[1]) + (red[2] + red[3]);
    }
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
      f4 a = {0, 0, 0, 0};
      for (int n = 0; n < N; ++n) a += sm[n] * load_tok4<BF16>(xb, (int64_t)n * D + d, f16);
      *reinterpret_cast<f4*>(p.P + ((int64_t)b * Q + q) * D + d) = (a - shift) * inv;
    }
    if (threadIdx.x == 0) {
      f4 rec = {mx, l, 0.f, 0.f};
      *reinterpret_cast<f4*>(p.ML + ((int64_t)b * Q + q) * 4) = rec;
    }
    __syncthreads();
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void ep_pool_bwd_generic_kernel(PoolParams p) {
  extern __shared__ float sm[];       // weights of one query row: N floats
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const char* xb = reinterpret_cast<const char*>(p.x) + EP_IMG_OFF(p, b) * (BF16 ? 2 : 4);
  const float* ts = p.tokstat ? p.tokstat + (int64_t)(p.index ? p.index[b] : b) * N * 2 : nullptr;
  float* red = sm + N;
  for (int q = 0; q < Q; ++q) {
    const float* g = p.dP + ((int64_t)b * Q + q) * D;
    const float* ml = p.ML + ((int64_t)b * Q + q) * 4;
    const float mx = ml[0], inv = 1.0f / ml[1], delta = ml[2];
    float gsum = 0.f;
    if (ts) {
      float t = 0.f;
      for (int d = threadIdx.x; d < D; d += 256) t += g[d];
      t = wave_sum(t);
      if (lane == 0) red[w] = t;
      __syncthreads();
      gsum = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    }
    for (int n = w; n < N; n += 4) {
      float s = 0.f;
      for (int d = lane * 4; d < D; d += 256) {
        f4 xv = load_tok4<BF16>(xb, (int64_t)n * D + d);
        f4 gv = *reinterpret_cast<const f4*>(g + d);
        s = fmaf(gv.x, xv.x, s); s = fmaf(gv.y, xv.y, s); s = fmaf(gv.z, xv.z, s); s = fmaf(gv.w, xv.w, s);
      }
      s = wave_sum(s);
      if (ts) s = ts[2 * n + 1] * (s - ts[2 * n] * gsum);                 // dA = dP . xhat_n
      if (p.dabias) s += p.dabias[((int64_t)b * Q + q) * N + n];
      if (lane == 0) {
        const float a = __builtin_amdgcn_exp2f((p.S[((int64_t)b * Q + q) * N + n] - mx) * LOG2E) * inv;
        const float ds = a * (s - delta);
        sm[n] = ds;
        if (p.dSout) p.dSout[((int64_t)b * Q + q) * N + n] = ds;
      }
    }
    __syncthreads();
    float shift = 0.f;
    if (ts) {
      float c = 0.f;
      for (int n = threadIdx.x; n < N; n += 256) { const float er = sm[n] * ts[2 * n + 1]; c = fmaf(er, ts[2 * n], c); sm[n] = er; }
      c = wave_sum(c);
      if (lane == 0) red[w] = c;
      __syncthreads();
      shift = (red[0] + red[1]) + (red[2] + red[3]);
    }
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
      f4 a = {0, 0, 0, 0};
      for (int n = 0; n < N; ++n) a += sm[n] * load_tok4<BF16>(xb, (int64_t)n * D + d);
      *reinterpret_cast<f4*>(p.Gpart + ((int64_t)b * Q + q) * D + d) = a - shift;
    }
    __syncthreads();
  }
}

// per-token LayerNorm statistics {mean, rstd} over D (biased variance, eps inside the root: torch.nn.LayerNorm);
// one wave per token, two passes over the row (the second one hits L2)
template <bool BF16>
__global__ __launch_bounds__(256) void ep_token_stats_kernel(const void* __restrict__ x, int64_t bstride, int N, int64_t rows,
                                                           int D, float eps, float* __restrict__ stats) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = (r / N) * bstride + (r % N) * (int64_t)D;          // first element of token r
  float s = 0.f;
  for (int d = lane * 4; d < D; d += 256) { const f4 v = load_tok4<BF16>(x, e0 + d); s += (v.x + v.y) + (v.z + v.w); }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const f4 v = load_tok4<BF16>(x, e0 + d) - mean;
    q = fmaf(v.x, v.x, q); q = fmaf(v.y, v.y, q); q = fmaf(v.z, v.z, q); q = fmaf(v.w, v.w, q);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { stats[r * 2] = mean; stats[r * 2 + 1] = rstd; }
}

// rows of up to 1280 values: the row stays in registers (CPL 16-byte chunks per lane) -- ONE read of the tokens; same
// operations in the same order as ep_token_stats_kernel, so the two give the same bits
template <bool BF16, int CPL>
__global__ __launch_bounds__(256) void ep_token_stats_reg_kernel(const void* __restrict__ x, int64_t bstride, int N, int64_t rows,
                                                               int D, float eps, float* __restrict__ stats) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = (r / N) * bstride + (r % N) * (int64_t)D;
  f4 v[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d = lane * 4 + 256 * k;
    v[k] = d < D ? load_tok4<BF16>(x, e0 + d) : f4{0.f, 0.f, 0.f, 0.f};
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k)
    if (lane * 4 + 256 * k < D) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k)
    if (lane * 4 + 256 * k < D) {
      const f4 c = v[k] - mean;
      q = fmaf(c.x, c.x, q); q = fmaf(c.y, c.y, q); q = fmaf(c.z, c.z, q); q = fmaf(c.w, c.w, q);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { stats[r * 2] = mean; stats[r * 2 + 1] = rstd; }
}

int token_stats(const void* x, int x_bf16, int64_t bstride, int B, int N, int D, float eps, float* stats, hipStream_t st) {
  const int64_t rows = (int64_t)B * N;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int cpl = (D + 255) / 256;
#define EP_TS(C_)                                                                                                          \
  if (cpl == C_) {                                                                                                         \
    if (x_bf16) hipLaunchKernelGGL((ep_token_stats_reg_kernel<true, C_>), dim3(grid), dim3(256), 0, st, x, bstride, N, rows, D, eps, stats); \
    else hipLaunchKernelGGL((ep_token_stats_reg_kernel<false, C_>), dim3(grid), dim3(256), 0, st, x, bstride, N, rows, D, eps, stats);       \
    EP_LAUNCH_CHECK("ep_token_stats_reg_kernel");                                                                          \
    return 0;                                                                                                              \
  }
  EP_TS(1) EP_TS(2) EP_TS(3) EP_TS(4) EP_TS(5)
#undef EP_TS
  if (x_bf16) hipLaunchKernelGGL(ep_token_stats_kernel<true>, dim3(grid), dim3(256), 0, st, x, bstride, N, rows, D, eps, stats);
  else hipLaunchKernelGGL(ep_token_stats_kernel<false>, dim3(grid), dim3(256), 0, st, x, bstride, N, rows, D, eps, stats);
  EP_LAUNCH_CHECK("ep_token_stats_kernel");
  return 0;
}

__global__ void ep_attention_kernel(const float* __restrict__ S, const float* __restrict__ ML,
                                    int rows, int N, float* __restrict__ A) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const float mx = ML[(int64_t)r * 4], inv = 1.0f / ML[(int64_t)r * 4 + 1];
  for (int n = threadIdx.x; n < N; n += blockDim.x)
    A[(int64_t)r * N + n] = __builtin_amdgcn_exp2f((S[(int64_t)r * N + n] - mx) * LOG2E) * inv;
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
StreamPlan stream_plan(int B, int N, int D, int Q) {
  StreamPlan c{};
  c.ok = false;
  (void)N;
  if (D % 64 != 0 || D > 1536 || Q < 1 || Q > 32) return c;
  c.kp = (D + 255) / 256;
  // queries per wave (QW) x waves per workgroup (NW) >= Q; 8 waves per CU (2 per SIMD)
  if (Q <= 4) { c.qw = 1; c.nw = 4; }               // waves without a query only feed the DMA ring
  else if (Q <= 8) { if (c.kp >= 5) { c.qw = 1; c.nw = 8; } else { c.qw = 2; c.nw = 4; } }   // wide rows: 1 query per wave
  else if (Q <= 16) { c.qw = 2; c.nw = 8; }
  else { c.qw = 4; c.nw = 8; }
  if (!stream_valid(c.qw, c.kp, c.nw)) return c;
  int wg_per_cu = stream_waves_per_cu(c.qw, c.kp, c.nw) / c.nw;
  if (const char* e = getenv("EP_POOL_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) wg_per_cu = v; }
  int grid = cu_count() * wg_per_cu;
  if (const char* e = getenv("EP_POOL_GRID")) { int v = atoi(e); if (v >= 1) grid = v; }   // diagnostics
  if (grid > B) grid = B;
  c.grid = grid;
  c.ok = true;
  return c;
}

// kernel selection: 0 = automatic (matrix-core kernel where supported, else the vector-ALU
// streaming kernel, else generic), 1 = generic only, 2 = skip the matrix-core kernel.
static int g_pool_mode = -1;
static int pool_mode() {
  if (g_pool_mode < 0) {
    const char* e = getenv("EP_POOL_MODE");
    g_pool_mode = e ? atoi(e) : 0;
  }
  return g_pool_mode;
}
int debug_force_generic(int mode) { int old = pool_mode(); g_pool_mode = mode; return old; }
static bool force_generic() { return pool_mode() == 1; }
static bool needs_generic(const PoolParams& p) { return p.sbias || p.dabias || p.dSout; }
// Measured on MI355X (tools/compare_modes.sh, bench.py): the vector-ALU kernel wins when it can keep
// three workgroups per CU (Q <= 8 and D <= 768) and for few queries (purely memory-bound: 6.2 TB/s at
// Q = 1); the matrix-core kernel wins for wider rows and more queries.
static bool use_mf(const PoolParams& p, bool bwd) {
  (void)bwd;
  if (needs_generic(p)) return false;
  if (p.tokstat || p.x_bf16 || pool_mode() != 0 || !mf_supported(p.D, p.Q, p.cls_bstride)) return false;
  const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  static int force_mf = -1;
  if (force_mf < 0) { const char* e = getenv("EP_POOL_MF_FORCE"); force_mf = e ? atoi(e) : 0; }
  if (!force_mf && c.ok && (p.Q <= 4 || stream_waves_per_cu(c.qw, c.kp, c.nw) == 12)) return false;   // vector-ALU kernel wins
  return true;
}
static int mf_grid(int B) { int g = cu_count(); return g < B ? g : B; }
// all-matrix-core kernel: forced with mode 3; chosen automatically where it measured fastest: more than 8 queries
// (16 heads at 256x768: 223 / 228 us against 250 / 266 us of the mixed kernel; at 256x1152: 332 us against 511 us of the
// vector-ALU kernel), and the backward at D = 1152, where the other two kernels run short of LDS / registers
// 17 .. 32 queries on fp32 tokens: both 16-query blocks against one read of the tokens (ep_pool_mm2.hip; EP_POOL_MM2=0: the
// round-4 dispatch -- forward in two chunks, backward on the vector-ALU kernel)
static bool use_mm2(const PoolParams& p, bool bwd) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_POOL_MM2"); on = e ? atoi(e) : 1; }
  if (!on || needs_generic(p) || p.tokstat || p.x_bf16 || (pool_mode() != 0 && pool_mode() != 3)) return false;
  return mm2_supported(p.D, p.Q, p.cls_bstride, bwd);
}
static bool use_mm(const PoolParams& p, bool bwd) {
  if (needs_generic(p)) return false;
  static int ln_mm = -1;                 // LayerNorm-of-tokens mode on the all-matrix-core kernel (EP_POOL_LN_MM=0: vector-ALU kernel)
  if (ln_mm < 0) { const char* e = getenv("EP_POOL_LN_MM"); ln_mm = e ? atoi(e) : 1; }
  if ((p.tokstat && !ln_mm) || p.x_bf16 || !mm_supported(p.D, p.Q, p.cls_bstride)) return false;
  if (pool_mode() == 3) return true;
  if (pool_mode() != 0) return false;
  return p.Q > 8 || (bwd && p.D == 1152 && p.Q >= 5);
}

// bf16-stored tokens: both contractions on the bf16 matrix cores (ep_pool_mb.hip), fp32 operand split into three bf16
// terms.  EP_POOL_MB=0 disables it (the vector-ALU kernels then widen the tokens on load).
static bool use_mb(const PoolParams& p) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_POOL_MB"); on = e ? atoi(e) : 1; }
  return on && p.x_bf16 == 1 && !needs_generic(p) && !p.tokstat && pool_mode() == 0 && mb_supported(p.D, p.Q, p.cls_bstride);
}

// bf16 tokens, 17 .. 32 queries: both 16-query blocks against one read of the tokens (second half of ep_pool_mb.hip;
// EP_POOL_MBQ=0: two 16-query launches as in round 4)
static bool use_mbq(const PoolParams& p) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_POOL_MBQ"); on = e ? atoi(e) : 1; }
  static int mb = -1;
  if (mb < 0) { const char* e = getenv("EP_POOL_MB"); mb = e ? atoi(e) : 1; }
  return on && mb && p.x_bf16 == 1 && !needs_generic(p) && !p.tokstat && pool_mode() == 0 && mbq_supported(p.D, p.Q, p.cls_bstride);
}

// what the vector-ALU streaming kernels can take (LayerNorm-of-tokens mode on bf16 tokens: where a tile's scores and
// statistics fit one 64-lane piece)
static bool stream_takes(const PoolParams& p) {
  static int ln = -1;
  if (ln < 0) { const char* e = getenv("EP_POOL_LN_STREAM"); ln = e ? atoi(e) : 1; }
  return !p.tokstat || (ln && (p.x_bf16 ? stream_ln_bf16_supported(p.D, p.Q) : stream_ln_supported(p.D, p.Q)));
}

// wide rows (D = 2048 / 4096): the row is split across the waves of a workgroup
static bool use_wide(const PoolParams& p) {
  return !needs_generic(p) && !p.tokstat && pool_mode() == 0 && p.x_bf16 != 2 && wide_supported(p.D, p.Q, p.cls_bstride, p.x_bf16);
}

// More queries than the fast kernel families take at this row width -- Q > 16 beyond D = 768, Q > 8 for the wide-row kernels:
// the reference's PUBLISHED rows all train with --ep_queries 32 (README.md:133-134), on 1024-, 1152- and 4096-wide tokens among
// others, and those shapes used to fall to the generic kernel (196 x 1024, Q = 32: 8.7 ms per pass, 18.6 ms per step).  The pass
// then runs in CHUNKS of queries on the fast family (PoolParams.Qs keeps the memory stride): the tokens are read once per
// chunk -- two reads at Q = 32 on the all-matrix-core kernel, four on the wide-row kernel -- instead of once at L2 speed.
// Returns the chunk size, 0 = no chunking.
static int query_chunk(const PoolParams& p, bool bwd) {
  if (needs_generic(p) || p.tokstat || force_generic() || pool_mode() != 0 || p.Q <= 8) return 0;
  if (use_wide(p) || use_mb(p) || use_mbq(p) || use_mm2(p, bwd) || use_mm(p, bwd) || use_mf(p, bwd)) return 0;
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_POOL_QCHUNK"); on = e ? atoi(e) : 1; }
  if (!on) return 0;
  if (p.x_bf16 && p.Q > 16) {            // bf16 tokens: two reads on the matrix-core pass beat one on the widening vector-ALU kernel
    PoolParams q = p;                    // (256 x 768, Q = 32: 500 / 454 us per pass on the vector-ALU kernel)
    const int n = (p.Q + 15) / 16, sz = (p.Q + n - 1) / n;     // even split: every chunk on the same (stride-aware) family
    q.Q = sz;
    PoolParams r = p;
    r.Q = p.Q - sz * (n - 1);
    if (use_mb(q) && use_mb(r)) return sz;
  }
  const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  // The vector-ALU kernel takes up to 32 queries in one pass, but beyond 16 its FORWARD is slower than two 16-query chunks on
  // the all-matrix-core kernel (256 x 768, Q = 32: 445 -> 385 us; the backward 407 -> 417 us, so it stays): round 4, default.
  // EP_POOL_QCHUNK=2: both directions in chunks; 3: neither.
  const bool prefer_chunks = p.Q > 16 && on != 3 && (on == 2 || !bwd);
  if (c.ok && stream_takes(p) && !prefer_chunks) return 0;
  // EVERY chunk -- the last, smaller one too -- must land on a family that honours the memory stride Qs (the vector-ALU and
  // the generic kernels do not): the queries are split evenly over ceil(Q / cap) chunks and both chunk sizes are checked
  // (round 5: Q = 24 at D = 512 used to run 16 + 8, and the 8-query remainder fell to the vector-ALU kernel, which wrote its
  // rows with stride 8 -- tests/test_gpu_bench_batch.py 33x77x512_q24)
  auto aware = [&](int qn) {
    PoolParams q = p;
    q.Q = qn;
    return use_wide(q) || use_mb(q) || use_mm(q, bwd) || use_mf(q, bwd);
  };
  for (int cap = 16; cap >= 8; cap -= 8) {
    if (p.Q <= cap) continue;
    const int n = (p.Q + cap - 1) / cap, sz = (p.Q + n - 1) / n, last = p.Q - sz * (n - 1);
    if (last >= 1 && aware(sz) && aware(last)) return sz;
  }
  return 0;
}

const char* pool_kernel_family(int B, int N, int D, int Q, int bwd, int x_bf16) {
  PoolParams p{};
  p.B = B; p.N = N; p.D = D; p.Q = Q; p.x_bf16 = x_bf16;
  if (x_bf16 == 2) return bwd ? "(none: fp16-stored tokens are forward only)" : (stream_plan(B, N, D, Q).ok && !force_generic() ? "ep_pool_fwd_kernel" : "ep_pool_fwd_generic_kernel");
  if (const int qc = query_chunk(p, bwd != 0)) p.Q = qc;          // (the family the chunks run on)
  if (use_wide(p)) {
    if (wideb_supported(p.D, p.Q, p.cls_bstride, p.x_bf16, bwd) && !p.tokstat) return bwd ? "ep_pool_wideb_bwd_kernel" : "ep_pool_wideb_fwd_kernel";
    return bwd ? "ep_pool_wide_bwd_kernel" : "ep_pool_wide_fwd_kernel";
  }
  if (use_mb(p)) return mb_kernel_name(D, bwd != 0);
  if (use_mbq(p)) return bwd ? "ep_pool_mbq_bwd_kernel" : "ep_pool_mbq_fwd_kernel";
  if (use_mm2(p, bwd != 0)) return bwd ? "ep_pool_mm2_bwd_kernel" : "ep_pool_mm2_fwd_kernel";
  if (use_mm(p, bwd != 0)) return bwd ? "ep_pool_mm_bwd_kernel" : "ep_pool_mm_fwd_kernel";
  if (use_mf(p, bwd != 0)) return bwd ? "ep_pool_mf_bwd_kernel" : "ep_pool_mf_fwd_kernel";
  if (stream_plan(B, N, D, Q).ok && !force_generic() && stream_takes(p)) return bwd ? "ep_pool_bwd_kernel" : "ep_pool_fwd_kernel";
  return bwd ? "ep_pool_bwd_generic_kernel" : "ep_pool_fwd_generic_kernel";
}

size_t pool_workspace_bytes(int B, int N, int D, int Q) {
  // Gpart: one (Q,D) partial per workgroup of the backward; the generic kernel uses one per
  // image, the streaming kernel one per resident workgroup (<= B).
  (void)N;
  return round_up((size_t)(2 * B + 16) * Q * D * sizeof(float), 256);
}

int pool_forward(const PoolParams& p0, hipStream_t st) {
  PoolParams p = p0;
  if (p.x_bf16 == 2) {
    // fp16-STORED tokens (ABI v24; what the reference's evaluate() hands the head under fp16 autocast, engine_finetune.py:131):
    // the vector-ALU streaming kernel widens them in its ring like bf16 (exact), the generic kernel on load; forward only
    EP_REQUIRE(!needs_generic(p) && !p.tokstat, EP_E_UNSUPPORTED, "fp16-stored tokens: plain shared-query forward only");
    const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
    if (c.ok && !force_generic()) return stream_launch(false, c, p, st);
    const size_t lds = (size_t)(p.N + 8) * sizeof(float);
    EP_REQUIRE(lds <= 64 * 1024, EP_E_UNSUPPORTED, "generic pooling kernel: N = %d tokens per image exceed its LDS row buffer (max 16376)", p.N);
    hipLaunchKernelGGL(ep_pool_fwd_generic_kernel<true>, dim3(p.B), dim3(256), lds, st, p);
    EP_LAUNCH_CHECK("ep_pool_fwd_generic_kernel");
    return 0;
  }
  if (const int qc = query_chunk(p, false)) {
    const int qs = p.Qs ? p.Qs : p.Q;
    for (int q0 = 0; q0 < p.Q; q0 += qc) {
      PoolParams q = p;
      q.Q = (p.Q - q0) < qc ? (p.Q - q0) : qc; q.Qs = qs;
      q.cls = p.cls + (int64_t)q0 * p.D; q.P = p.P + (int64_t)q0 * p.D; q.S = p.S + (int64_t)q0 * p.N; q.ML = p.ML + (int64_t)q0 * 4;
      EP_TRY(pool_forward(q, st));
    }
    return 0;
  }
  // per-image query rows with score extras (CLIP): every token read once for all heads instead of once per head
  if (needs_generic(p) && !force_generic() && imgqf_supported(p)) return imgqf_forward(p, st);
  if (use_wide(p)) return wide_launch(false, p, wide_grid(p.D, p.B, p.x_bf16), st);
  if (use_mb(p)) {
    if (const char* e = getenv("EP_POOL_ABLATE")) p.ablate = atoi(e);
    return mb_launch(false, p, mb_grid(p.D, p.B), st);
  }
  if (use_mbq(p)) return mbq_launch(false, p, mbq_grid(p.B), st);
  if (use_mm2(p, false)) return mm2_launch(false, p, mf_grid(p.B), st);
  if (use_mm(p, false)) return mm_launch(false, p, mf_grid(p.B), st);
  if (use_mf(p, false)) return mf_launch(false, p, mf_grid(p.B), st);
  // (the kernels below index P / S / ML with Q as the per-image row stride)
  EP_REQUIRE(!p.Qs || p.Qs == p.Q, EP_E_UNSUPPORTED, "pool_forward: a %d-query chunk of %d has no stride-aware kernel at D=%d", p.Q, p.Qs, p.D);
  StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  if (c.ok && !force_generic() && !needs_generic(p) && stream_takes(p)) {
    if (const char* e = getenv("EP_POOL_ABLATE")) p.ablate = atoi(e);
    return stream_launch(false, c, p, st);
  }
  const size_t lds = (size_t)(p.N + 8) * sizeof(float);
  EP_REQUIRE(lds <= 64 * 1024, EP_E_UNSUPPORTED, "generic pooling kernel: N = %d tokens per image exceed its LDS row buffer (max 16376)", p.N);
  if (p.x_bf16) hipLaunchKernelGGL(ep_pool_fwd_generic_kernel<true>, dim3(p.B), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(ep_pool_fwd_generic_kernel<false>, dim3(p.B), dim3(256), lds, st, p);
  EP_LAUNCH_CHECK("ep_pool_fwd_generic_kernel");
  return 0;
}

// The vector-ALU streaming kernel with 4-wave workgroups can carry side work (256-thread tasks).
bool pool_backward_takes_side(const PoolParams& p) {
  static int allow = -1;
  if (allow < 0) { const char* e = getenv("EP_POOL_SIDE"); allow = e ? atoi(e) : 1; }
  // (LayerNorm-of-tokens mode of the vector-ALU kernel: since round 4 -- CaiT's five weight gradients in its pass, 1.167 ->
  // 1.158 ms against their early start on the aux stream; EP_POOL_SIDE_LN=0 keeps them there)
  static int allow_ln = -1;
  if (allow_ln < 0) { const char* e = getenv("EP_POOL_SIDE_LN"); allow_ln = e ? atoi(e) : 1; }
  if (!allow || needs_generic(p) || (p.tokstat && !allow_ln) || use_wide(p) || force_generic()) return false;
  if (use_mb(p)) return mb_takes_side(p.D);          // bf16 tokens: the two-workgroup matrix-core pass carries them too
  if (use_mbq(p)) return false;
  if (use_mm2(p, true) || use_mm(p, true) || use_mf(p, true)) return false;
  const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  return c.ok && c.nw == 4;
}

// The same kernel can compute the softmax-correction rows itself (PoolParams.dyv / yv / Dv): one ring item per image
// holds dy[b] | y[b] (Dv floats each, 1-KiB pieces), every wave reduces the slices of its queries.
bool pool_backward_takes_delta(const PoolParams& p, int Dv) {
  static int allow = -1;
  if (allow < 0) { const char* e = getenv("EP_POOL_DELTA"); allow = e ? atoi(e) : 1; }
  if (!allow || needs_generic(p) || p.tokstat || use_wide(p) || force_generic()) return false;
  // a chunked second pass (pool_backward asks query_chunk first) runs one launch per chunk with the chunk's Q: no launch sees
  // the whole dy . y row -- the delta rows then come from ep_delta_kernel (the two predicates must agree)
  if (query_chunk(p, true) != 0) return false;
  if (use_mb(p)) return mb_takes_delta(p.D, p.Q, Dv);
  if (use_mbq(p)) return mbq_takes_delta(p.D, p.Q, Dv);
  if (use_mm2(p, true) || use_mm(p, true) || use_mf(p, true)) return false;
  const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  if (!c.ok || !stream_takes(p) || Dv <= 0 || Dv % (4 * p.Q) != 0) return false;
  // ring slot of the kernel that will run = (tokens per tile) * D * (bytes per stored element).  A bf16 tile holds TWICE
  // the tokens (ep_pool_stream.hip: TT = Cfg::TT * (BF16 ? 2 : 1)) at half the bytes each, so the slot is
  // stream_tt * D * 4 bytes for both storage types (tests/test_gpu_step_folds.py: bf16 at D = 192 / 640)
  const size_t tt = (size_t)stream_tt(c.qw, c.kp, c.nw) * (p.x_bf16 ? 2 : 1);
  const size_t slot = tt * p.D * (p.x_bf16 ? 2 : 4);
  return 2 * (((size_t)Dv * 4 + 1023) / 1024) * 1024 <= slot;
}

// In-pass contractions (ep_inpass.h): both passes on the 4-wave / 2-queries-per-wave streaming kernel, Q = 8, D = 256 kp,
// projection width = D, whole row blocks of 32 images, pooling grid a multiple of 32 (a row block's producers then sit in
// one aligned group of 32 workgroups).
int pool_inpass_mask(const PoolParams& p, int Dv) {
  static int want = -1;
  if (want < 0) { const char* e = getenv("EP_INPASS"); want = e ? atoi(e) : 2; }   // default: dP inside the second pass
  if (!want || needs_generic(p) || p.tokstat || use_wide(p) || force_generic()) return 0;
  if (use_mb(p)) {
    // bf16-stored tokens: dP inside the second matrix-core pass (ep_pool_mb.hip); same shape rules as below
    static int mb_ip = -1;
    if (mb_ip < 0) { const char* e = getenv("EP_INPASS_MB"); mb_ip = e ? atoi(e) : 1; }
    if (!(want & 2) || !mb_ip || Dv != p.D || p.B % 32 != 0 || mb_grid(p.D, p.B) % 32 != 0 || p.cls_bstride != 0 ||
        (int64_t)p.B * p.Q * p.D * 4 >= (int64_t)0x7fffffff || !pool_backward_takes_side(p) || !mb_takes_inpass_dp(p))
      return 0;
    return 2;
  }
  if (use_mm(p, true) || use_mf(p, true) || use_mm(p, false) || use_mf(p, false) || !stream_takes(p)) return 0;
  const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  if (!c.ok || c.nw != 4 || c.qw != 2 || c.kp > 3 || p.Q != 8 || p.D != 256 * c.kp || Dv != p.D || p.B % 32 != 0 ||
      c.grid % 32 != 0 || p.cls_bstride != 0 || (int64_t)p.B * p.Q * p.D * 4 >= (int64_t)0x7fffffff)   // (32-bit buffer offsets)
    return 0;
  static int bwd_grid = -1;           // (diagnostic grid override of the second pass, see pool_backward)
  if (bwd_grid < 0) { const char* e = getenv("EP_POOL_BWD_GRID"); bwd_grid = e ? atoi(e) : 0; }
  if (bwd_grid > 0 && bwd_grid % 32 != 0) return 0;
  // Progress of the hand-off needs every pooling workgroup of the grid resident at once (a workgroup waits on row blocks
  // that later-indexed workgroups of the same launch produce): ask the runtime how many workgroups of each pass -- with
  // the tasks' LDS -- a CU holds, and keep the contractions as launches of their own when the grid is larger than that.
  // (A chip shared with another process or a CU mask can still starve a launch: the waits are bounded and a give-up stops
  // the training loop through the optimizer's abort flag, ep_api.hip.)
  {
    static int res[2][4][2] = {};      // [pass][kp][bf16]: 0 unknown, else blocks per CU + 1 (0 -> "cannot tell" is stored as INT_MAX)
    for (int bwd = 0; bwd < 2; ++bwd) {
      if (!((want >> bwd) & 1)) continue;
      int& r = res[bwd][c.kp][p.x_bf16 ? 1 : 0];
      if (r == 0) {
        PoolParams q = p;
        int dummy = 0;
        if (bwd) { q.ip_dy = reinterpret_cast<const float*>(&dummy); } else { q.ip_ycnt = &dummy; }
        SideTasks sd{};
        sd.total = 1;
        const int nb = stream_resident_blocks_per_cu(bwd != 0, c, q, bwd ? &sd : nullptr);
        r = nb < 0 ? 0x7fffffff : nb + 1;
      }
      if (r != 0x7fffffff && (int64_t)(r - 1) * cu_count() < c.grid) return 0;
    }
  }
  int m = want & 3;
  // ticketed second pass: at least 20 token tiles per image (its task point and the row-block check sit inside an image)
  // and at least 64 pooling workgroups (a row block's 32 tickets are then drawn by workgroups that never wait on it)
  const int tt = stream_tt(c.qw, c.kp, c.nw) * (p.x_bf16 ? 2 : 1);
  if ((want & 4) && (p.N + tt - 1) / tt >= 20 && c.grid >= 64) m |= 4 | 2;
  return m;
}

StreamGridInfo pool_stream_grid(const PoolParams& p) {
  const StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  StreamGridInfo g{c.grid, 1, 0};
  if (c.ok && c.grid > 0) {
    g.rounds = (p.B + c.grid - 1) / c.grid;
    g.helpers = c.grid - (p.B - (g.rounds - 1) * c.grid);
  }
  return g;
}

int pool_backward(const PoolParams& p0, float* dcls, int accumulate, hipStream_t st, const SideTasks* side,
                  DeferredReduce* defer) {
  PoolParams p = p0;
  EP_REQUIRE(p.x_bf16 != 2, EP_E_UNSUPPORTED, "fp16-stored tokens are a forward / evaluation storage type (widen them for training)");
  if (const int qc = query_chunk(p, true)) {
    EP_REQUIRE(!side || side->total == 0, EP_E_UNSUPPORTED, "pool_backward: side tasks with a chunked pass");
    const int qs = p.Qs ? p.Qs : p.Q;
    for (int q0 = 0; q0 < p.Q; q0 += qc) {
      PoolParams q = p;
      q.Q = (p.Q - q0) < qc ? (p.Q - q0) : qc; q.Qs = qs;
      q.dP = p.dP + (int64_t)q0 * p.D; q.S = p.S + (int64_t)q0 * p.N; q.ML = p.ML + (int64_t)q0 * 4;
      EP_TRY(pool_backward(q, dcls + (int64_t)q0 * p.D, accumulate, st, nullptr, nullptr));
    }
    if (defer) defer->stage = nullptr;
    return 0;
  }
  StreamPlan c = stream_plan(p.B, p.N, p.D, p.Q);
  int nparts;
  EP_REQUIRE(!side || side->total == 0 || pool_backward_takes_side(p), EP_E_UNSUPPORTED,
             "pool_backward: side tasks need the 4-wave streaming kernel or the two-workgroup bf16 matrix-core pass");
  if (use_wide(p)) {
    nparts = wide_grid(p.D, p.B, p.x_bf16);
    EP_TRY(wide_launch(true, p, nparts, st));
  } else if (use_mb(p)) {
    nparts = mb_grid(p.D, p.B);
    EP_TRY(mb_launch(true, p, nparts, st, side));
  } else if (use_mbq(p)) {
    nparts = mbq_grid(p.B);
    EP_TRY(mbq_launch(true, p, nparts, st));
  } else if (use_mm2(p, true)) {
    nparts = mf_grid(p.B);
    EP_TRY(mm2_launch(true, p, nparts, st));
  } else if (use_mm(p, true)) {
    nparts = mf_grid(p.B);
    EP_TRY(mm_launch(true, p, nparts, st));
  } else if (use_mf(p, true)) {
    const int grid = mf_grid(p.B);
    nparts = 2 * grid;                         // one partial per token half of every workgroup
    EP_TRY(mf_launch(true, p, grid, st));
  } else if (p.Qs && p.Qs != p.Q) {
    // (the kernels below index S / ML / dP with Q as the per-image row stride)
    set_error("pool_backward: a %d-query chunk of %d has no stride-aware kernel at D=%d", p.Q, p.Qs, p.D);
    return EP_E_UNSUPPORTED;
  } else if (p.tick) {
    // ticketed form (ep_pool_bwd2.hip): one gradient partial per IMAGE; `first` pooling workgroups in front of the side tasks
    static int first_env = -1;
    if (first_env < 0) { const char* e = getenv("EP_BWD2_FIRST"); first_env = e ? atoi(e) : 0; }
    int first = (side && side->total > 0) ? (c.grid * 2 / 3) / 32 * 32 : c.grid;
    if (first_env > 0) first = first_env < c.grid ? first_env : c.grid;
    p.tick_base = first;
    EP_TRY(bwd2_launch(p, c.grid, first, st, side));
    nparts = p.B;
  } else if (c.ok && !force_generic() && !needs_generic(p) && stream_takes(p)) {
    static int bwd_grid = -1;           // diagnostics: pooling workgroups of the SECOND pass only (e.g. 2 per CU, the third slot left to the side work)
    if (bwd_grid < 0) { const char* e = getenv("EP_POOL_BWD_GRID"); bwd_grid = e ? atoi(e) : 0; }
    if (bwd_grid > 0 && side && side->total > 0) c.grid = bwd_grid < p.B ? bwd_grid : p.B;
    EP_TRY(stream_launch(true, c, p, st, side));
    nparts = c.grid;
  } else {
    const size_t lds = (size_t)(p.N + 8) * sizeof(float);
    EP_REQUIRE(lds <= 64 * 1024, EP_E_UNSUPPORTED, "generic pooling kernel: N = %d tokens per image exceed its LDS row buffer (max 16376)", p.N);
    if (p.x_bf16) hipLaunchKernelGGL(ep_pool_bwd_generic_kernel<true>, dim3(p.B), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(ep_pool_bwd_generic_kernel<false>, dim3(p.B), dim3(256), lds, st, p);
    EP_LAUNCH_CHECK("ep_pool_bwd_generic_kernel");
    nparts = p.B;
  }
  return reduce_partials(p.Gpart, nparts, p.Q * p.D, p.scale, accumulate, dcls, p.Gpart + (int64_t)nparts * p.Q * p.D, st, defer);
}

// The generic kernel writes one (Q,D) partial PER IMAGE: with per-image query rows those partials are the result.
__global__ void ep_scale_copy_kernel(const float* __restrict__ src, int64_t n, float scale, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * scale;
}
int pool_backward_per_image(const PoolParams& p0, float* dq, hipStream_t st) {
  PoolParams p = p0;
  if (!force_generic() && imgqf_supported(p)) return imgqf_backward(p, dq, st);
  p.Gpart = dq;
  const size_t lds = (size_t)(p.N + 8) * sizeof(float);
  EP_REQUIRE(lds <= 64 * 1024, EP_E_UNSUPPORTED, "generic pooling kernel: N = %d tokens per image exceed its LDS row buffer (max 16376)", p.N);
  if (p.x_bf16) hipLaunchKernelGGL(ep_pool_bwd_generic_kernel<true>, dim3(p.B), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(ep_pool_bwd_generic_kernel<false>, dim3(p.B), dim3(256), lds, st, p);
  if (p.scale != 1.0f) {
    const int64_t n = (int64_t)p.B * p.Q * p.D;
    hipLaunchKernelGGL(ep_scale_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dq, n, p.scale, dq);
  }
  EP_LAUNCH_CHECK("ep_pool_bwd_generic_kernel (per image)");
  return 0;
}

// out[n] (+)= scale * sum_i parts[i][n] in two deterministic stages for many parts:
// nparts -> 16 group sums (`stage`, 16*n floats) -> out
int reduce_partials(const float* parts, int nparts, int n, float scale, int accumulate, float* out, float* stage,
                    hipStream_t st, DeferredReduce* defer) {
  const int gx = (n / 4 + 63) / 64;
  if (defer) defer->stage = nullptr;
  if (nparts > 32) {
    hipLaunchKernelGGL(ep_reduce_partials_kernel, dim3(gx, 16), dim3(256), 0, st, parts, nparts, n, 1.0f, 0, stage);
    if (defer && n % 4 == 0) { *defer = DeferredReduce{stage, out, n, scale, accumulate}; }
    else hipLaunchKernelGGL(ep_reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, stage, 16, n, scale, accumulate, out);
  } else {
    hipLaunchKernelGGL(ep_reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, parts, nparts, n, scale, accumulate, out);
  }
  EP_LAUNCH_CHECK("ep_reduce_partials_kernel");
  return 0;
}

int attention_from_scores(const float* S, const float* ML, int rows, int N, float* A, hipStream_t st) {
  hipLaunchKernelGGL(ep_attention_kernel, dim3(rows), dim3(256), 0, st, S, ML, rows, N, A);
  EP_LAUNCH_CHECK("ep_attention_kernel");
  return 0;
}

}  // namespace ep
