// Token passes with per-image query rows over channel slices (see ep_pool_imgq.h).
#include "ep_pool_imgq.h"

namespace ep {

constexpr float IQ_LOG2E = 1.4426950408889634f;

template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 8 || G == 16 || G == 64, "lane group of 8, 16 or 64");
  if (G == 64) return wave_sum(v);
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);            // row_half_mirror: the 8 lanes of a half row
  if (G == 16) v += dpp_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float dot4(f4 a, f4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ float sum4(f4 a) { return (a.x + a.y) + (a.z + a.w); }

// per-lane record exchanged through LDS when the token waves of a head are merged
template <int CPL>
struct IqRec { static constexpr int FLOATS = 4 + 4 * CPL; };

template <int G, int CPL, int TB, bool BF16, bool BWD>
__global__ void ep_imgq_kernel(ImgqParams p, int HW, int TW) {
  extern __shared__ __attribute__((aligned(16))) float iq_lds[];
  constexpr int GPW = 64 / G;                        // heads per wave
  constexpr int REC = IqRec<CPL>::FLOATS;
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int hw = w % HW, tw = w / HW;
  const int D = p.D, N = p.N, H = p.H, Dh = D / H;
  const int gl = lane / G, sl = lane % G;
  const int head_raw = hw * GPW + gl;
  const bool head_ok = head_raw < H;
  const int head = head_ok ? head_raw : H - 1;

  int ch[CPL];
  bool cv[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = 4 * (sl + G * j);
    cv[j] = head_ok && c < Dh;
    ch[j] = head * Dh + (c < Dh ? c : 0);
  }
  const int64_t img = (int64_t)(p.index ? p.index[b] : b);
  const void* xb = BF16 ? static_cast<const void*>(static_cast<const uint16_t*>(p.x) + img * p.x_bstride)
                        : static_cast<const void*>(static_cast<const float*>(p.x) + img * p.x_bstride);
  const float* ts = p.tokstat ? p.tokstat + img * N * 2 : nullptr;
  const bool pool_ln = p.pool_ln != 0;

  f4 uq[CPL], gq[BWD ? CPL : 1], acc[CPL];
  float usum = 0.f, gsum = 0.f, delta = 0.f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    uq[j] = cv[j] ? *reinterpret_cast<const f4*>(p.u + (int64_t)b * D + ch[j]) : z;
    usum += sum4(uq[j]);
    acc[j] = z;
    if constexpr (BWD) {
      gq[j] = cv[j] ? *reinterpret_cast<const f4*>(p.dP + (int64_t)b * D + ch[j]) : z;
      gsum += sum4(gq[j]);
      const f4 pq = cv[j] ? *reinterpret_cast<const f4*>(p.P + (int64_t)b * D + ch[j]) : z;
      delta += dot4(gq[j], pq);
    }
  }
  usum = group_sum<G>(usum);
  float m = -INFINITY, l = 0.f, cacc = 0.f, invl = 0.f;
  if constexpr (BWD) {
    gsum = group_sum<G>(gsum);
    delta = group_sum<G>(delta);
    m = p.ML[((int64_t)b * H + head) * 2];
    invl = 1.0f / p.ML[((int64_t)b * H + head) * 2 + 1];
  }

  for (int n0 = tw * TB; n0 < N; n0 += TW * TB) {
    f4 xv[TB][CPL];
    float mean[TB], rstd[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int n = (n0 + t) < N ? (n0 + t) : N - 1;
#pragma unroll
      for (int j = 0; j < CPL; ++j) xv[t][j] = load_tok4<BF16>(xb, (int64_t)n * D + ch[j]);
      mean[t] = ts ? ts[2 * n] : 0.f;
      rstd[t] = ts ? ts[2 * n + 1] : 1.f;
    }
    float s[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) d += dot4(uq[j], xv[t][j]);
      d = group_sum<G>(d);
      s[t] = rstd[t] * (d - mean[t] * usum);          // (mean 0, rstd 1 without token statistics)
    }
    if constexpr (!BWD) {
      float bm = -INFINITY;
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        if (n0 + t >= N) s[t] = -INFINITY;
        bm = fmaxf(bm, s[t]);
      }
      const float mn = fmaxf(m, bm);                  // finite: token n0 is valid
      const float corr = __builtin_amdgcn_exp2f((m - mn) * IQ_LOG2E);   // m = -inf -> 0
      l *= corr; cacc *= corr;
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc[j] *= corr;
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const float pt = __builtin_amdgcn_exp2f((s[t] - mn) * IQ_LOG2E);
        l += pt;
        const float wt = pool_ln ? pt * rstd[t] : pt;
        cacc = fmaf(wt, mean[t], cacc);
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] += wt * xv[t][j];
      }
      m = mn;
    } else {
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) e += dot4(gq[j], xv[t][j]);
        e = group_sum<G>(e);
        const float dA = pool_ln ? rstd[t] * (e - mean[t] * gsum) : e;
        const float a = __builtin_amdgcn_exp2f((s[t] - m) * IQ_LOG2E) * invl;
        float dS = a * (dA - delta);
        if (n0 + t >= N) dS = 0.f;
        const float wt = dS * rstd[t];                 // dS k_n with k = xhat (or x: rstd 1, mean 0)
        cacc = fmaf(wt, mean[t], cacc);
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] += wt * xv[t][j];
      }
    }
  }

  // ---- merge the TW token waves of every head (fixed order) and store ----
  if (TW > 1) {
    float* rec = iq_lds + ((size_t)w * 64 + lane) * REC;
    rec[0] = m; rec[1] = l; rec[2] = cacc;
#pragma unroll
    for (int j = 0; j < CPL; ++j) *reinterpret_cast<f4*>(rec + 4 + 4 * j) = acc[j];
    __syncthreads();
    if (tw != 0) return;
    for (int i = 1; i < TW; ++i) {
      const float* o = iq_lds + ((size_t)(i * HW + hw) * 64 + lane) * REC;
      if constexpr (!BWD) {
        const float mi = o[0];
        if (mi > -INFINITY) {
          const float mn = fmaxf(m, mi);
          const float f0 = __builtin_amdgcn_exp2f((m - mn) * IQ_LOG2E), fi = __builtin_amdgcn_exp2f((mi - mn) * IQ_LOG2E);
          l = l * f0 + o[1] * fi;
          cacc = cacc * f0 + o[2] * fi;
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[j] = acc[j] * f0 + *reinterpret_cast<const f4*>(o + 4 + 4 * j) * fi;
          m = mn;
        }
      } else {
        cacc += o[2];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] += *reinterpret_cast<const f4*>(o + 4 + 4 * j);
      }
    }
  }
  if constexpr (!BWD) {
    const float inv = 1.0f / l;
    const float shift = pool_ln ? cacc : 0.f;          // sum_n A rstd_n mean_n
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (cv[j]) *reinterpret_cast<f4*>(p.P + (int64_t)b * D + ch[j]) = (acc[j] - shift) * inv;
    if (head_ok && sl == 0) {
      p.ML[((int64_t)b * H + head) * 2] = m;
      p.ML[((int64_t)b * H + head) * 2 + 1] = l;
    }
  } else {
    const float shift = ts ? cacc : 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (cv[j]) *reinterpret_cast<f4*>(p.du + (int64_t)b * D + ch[j]) = acc[j] - shift;
  }
}

struct IqPlan { int G, CPL, TB, HW, TW; bool ok; };

static IqPlan iq_plan(int D, int H) {
  IqPlan c{};
  c.ok = false;
  if (H < 1 || D % H != 0 || (D / H) % 4 != 0) return c;
  const int chunks = D / H / 4;
  if (chunks >= 64 || H == 1) { c.G = 64; c.CPL = (chunks + 63) / 64; }
  else if (chunks % 16 == 0) { c.G = 16; c.CPL = chunks / 16; }
  else if (chunks % 8 == 0) { c.G = 8; c.CPL = chunks / 8; }
  else return c;
  // instantiated chunk counts: 1..5 for every group width, 8 and 16 for whole-wave groups (D up to 4096 with one head)
  if (c.CPL > 5) {
    if (c.G != 64 || c.CPL > 16) return c;
    c.CPL = c.CPL <= 8 ? 8 : 16;
  }
  c.TB = c.CPL <= 3 ? 4 : (c.CPL <= 5 ? 2 : 1);
  c.HW = (H + 64 / c.G - 1) / (64 / c.G);
  if (c.HW > 16) return c;
  c.TW = c.HW >= 4 ? 1 : (c.HW == 3 ? 2 : 4 / c.HW);
  if (c.CPL >= 8 && c.TW > 2) c.TW = 2;
  c.ok = true;
  return c;
}

template <int G, int CPL, int TB>
static int iq_launch(const ImgqParams& p, const IqPlan& c, bool bwd, hipStream_t st) {
  const dim3 grid(p.B), block(64 * c.HW * c.TW);
  const size_t lds = c.TW > 1 ? (size_t)c.HW * c.TW * 64 * IqRec<CPL>::FLOATS * sizeof(float) : 0;
  if (p.x_bf16) {
    if (bwd) hipLaunchKernelGGL((ep_imgq_kernel<G, CPL, TB, true, true>), grid, block, lds, st, p, c.HW, c.TW);
    else hipLaunchKernelGGL((ep_imgq_kernel<G, CPL, TB, true, false>), grid, block, lds, st, p, c.HW, c.TW);
  } else {
    if (bwd) hipLaunchKernelGGL((ep_imgq_kernel<G, CPL, TB, false, true>), grid, block, lds, st, p, c.HW, c.TW);
    else hipLaunchKernelGGL((ep_imgq_kernel<G, CPL, TB, false, false>), grid, block, lds, st, p, c.HW, c.TW);
  }
  EP_LAUNCH_CHECK(bwd ? "ep_imgq_kernel (backward)" : "ep_imgq_kernel (forward)");
  return 0;
}

static int iq_dispatch(const ImgqParams& p, bool bwd, hipStream_t st) {
  const IqPlan c = iq_plan(p.D, p.H);
  EP_REQUIRE(c.ok, EP_E_UNSUPPORTED, "per-image-query token pass: D=%d with %d heads is not supported "
             "(head width must be a multiple of 32, at most 16 head-waves)", p.D, p.H);
  EP_REQUIRE(!p.pool_ln || p.tokstat, EP_E_ARG, "per-image-query token pass: pooling normalised tokens needs token statistics");
#define EP_IQ(G_, C_, T_) if (c.G == G_ && c.CPL == C_) return iq_launch<G_, C_, T_>(p, c, bwd, st);
  EP_IQ(64, 1, 4) EP_IQ(64, 2, 4) EP_IQ(64, 3, 4) EP_IQ(64, 4, 2) EP_IQ(64, 5, 2) EP_IQ(64, 8, 1) EP_IQ(64, 16, 1)
  EP_IQ(16, 1, 4) EP_IQ(16, 2, 4) EP_IQ(16, 3, 4)
  EP_IQ(8, 1, 4) EP_IQ(8, 3, 4) EP_IQ(8, 5, 2)
#undef EP_IQ
  set_error("per-image-query token pass: no kernel for group %d x %d chunks", c.G, c.CPL);
  return EP_E_UNSUPPORTED;
}

bool imgq_supported(int D, int H) { return iq_plan(D, H).ok; }
int imgq_forward(const ImgqParams& p, hipStream_t st) { return iq_dispatch(p, false, st); }
int imgq_backward(const ImgqParams& p, hipStream_t st) { return iq_dispatch(p, true, st); }

// =============================================================================================
// FULL-WIDTH per-image query rows (CLIP attention pooling: the query is the image's own mean row, H = 4 heads that each
// score and pool the WHOLE token row): PoolParams semantics of the generic kernels -- per-image rows p.cls (+ cls_bstride),
// optional LayerNorm-of-tokens mode, additive score bias, explicit scores S; backward with the stored scores, an additive
// dA term, explicit dS and PER-IMAGE query gradients -- but every token is read from HBM once for all heads.
// One 4-wave workgroup per image; a wave walks its tokens TB at a time, a lane owns CPL 16-byte chunks of the row for ALL
// heads (registers: HQ x CPL query / dP chunks and as many accumulators), a score is a lane-local dot product plus one
// full-wave reduction.  The four waves are merged through LDS head by head at the end of the image (fixed order).
// =============================================================================================
constexpr int IQF_NW = 4;

template <int CPL, int HQ, int TB, bool BF16, bool BWD>
__global__ __launch_bounds__(IQF_NW * 64) void ep_imgqf_kernel(PoolParams p, float* __restrict__ dq) {
  constexpr int REC = 4 + 4 * CPL;
  __shared__ __attribute__((aligned(16))) float rec_lds[IQF_NW * 64 * REC];
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  int ch[CPL];
  bool cv[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = 4 * (lane + 64 * j);
    cv[j] = c < D;
    ch[j] = cv[j] ? c : 0;
  }
  const int64_t img = (int64_t)(p.index ? p.index[b] : b);
  const void* xb = BF16 ? static_cast<const void*>(reinterpret_cast<const uint16_t*>(p.x) + img * p.x_bstride)
                        : static_cast<const void*>(p.x + img * p.x_bstride);
  const float* ts = p.tokstat ? p.tokstat + img * N * 2 : nullptr;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};

  f4 qv[HQ][CPL], acc[HQ][CPL];                      // forward: scaled query rows; backward: dP rows
  float qsum[HQ], m[HQ], l[HQ], cacc[HQ], delta[HQ];
#pragma unroll
  for (int h = 0; h < HQ; ++h) {
    const bool hv = h < Q;
    const float* row = BWD ? p.dP + ((int64_t)b * Q + (hv ? h : 0)) * D
                           : p.cls + (int64_t)b * p.cls_bstride + (int64_t)(hv ? h : 0) * D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      f4 v = (hv && cv[j]) ? *reinterpret_cast<const f4*>(row + ch[j]) : z4;
      if (!BWD) v = v * p.scale;
      qv[h][j] = v;
      acc[h][j] = z4;
      s += sum4(v);
    }
    qsum[h] = wave_sum(s);
    cacc[h] = 0.f;
    if (BWD) {
      const float* ml = p.ML + ((int64_t)b * Q + (hv ? h : 0)) * 4;
      m[h] = ml[0]; l[h] = 1.0f / ml[1]; delta[h] = ml[2];
    } else {
      m[h] = -INFINITY; l[h] = 0.f; delta[h] = 0.f;
    }
  }

  // Per-iteration scalars ride in lanes: lane i < HQ*TB owns (head i / TB, token i % TB) of the iteration -- its score bias
  // (forward) or stored score and dA term (backward), and the score / dS it writes back; lane i < 2*TB owns the i-th float of
  // the iteration's {mean, rstd} pairs.  v_readlane hands them to the whole wave.
  const int eh = lane / TB, et = lane % TB;
  const bool elane = eh < HQ && eh < Q;
  const int64_t erow = ((int64_t)b * Q + (elane ? eh : 0)) * N;
  auto rl = [](float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); };
  auto load = [&](int n0, f4 (&xv)[TB][CPL], float& tsv, float& e0, float& e1) {
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int n = (n0 + t) < N ? (n0 + t) : N - 1;
#pragma unroll
      for (int j = 0; j < CPL; ++j) xv[t][j] = cv[j] ? load_tok4<BF16>(xb, (int64_t)n * D + ch[j]) : z4;
    }
    {
      int e = 2 * n0 + lane; e = e < 2 * N ? e : 2 * N - 1;
      tsv = (ts && lane < 2 * TB) ? ts[e] : ((lane & 1) ? 1.f : 0.f);
    }
    const bool ev = elane && (n0 + et) < N;
    const int64_t ei = erow + (ev ? n0 + et : 0);
    if (!BWD) { e0 = (ev && p.sbias) ? p.sbias[ei] : 0.f; e1 = 0.f; }
    else { e0 = ev ? p.S[ei] : 0.f; e1 = (ev && p.dabias) ? p.dabias[ei] : 0.f; }
  };
  auto compute = [&](int n0, const f4 (&xv)[TB][CPL], float tsv, float e0, float e1) {
    float mean[TB], rstd[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) { mean[t] = rl(tsv, 2 * t); rstd[t] = rl(tsv, 2 * t + 1); }
    float outv = 0.f;                                  // this lane's score / dS of the iteration
#pragma unroll
    for (int h = 0; h < HQ; ++h) {
      if (h >= Q) continue;
      float s[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) d += dot4(qv[h][j], xv[t][j]);
        d = wave_sum(d);
        s[t] = rstd[t] * (d - mean[t] * qsum[h]);      // (mean 0, rstd 1 without token statistics)
      }
      if constexpr (!BWD) {
        float bm = -INFINITY;
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          s[t] += rl(e0, h * TB + t);
          outv = lane == h * TB + t ? s[t] : outv;
          if (n0 + t >= N) s[t] = -INFINITY;
          bm = fmaxf(bm, s[t]);
        }
        const float mn = fmaxf(m[h], bm);               // finite: token n0 is valid
        const float corr = __builtin_amdgcn_exp2f((m[h] - mn) * IQ_LOG2E);   // m = -inf -> 0
        l[h] *= corr; cacc[h] *= corr;
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[h][j] *= corr;
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const float pt = __builtin_amdgcn_exp2f((s[t] - mn) * IQ_LOG2E);
          l[h] += pt;
          const float wt = pt * rstd[t];
          cacc[h] = fmaf(wt, mean[t], cacc[h]);
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[h][j] += wt * xv[t][j];
        }
        m[h] = mn;
      } else {
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const float dA = s[t] + rl(e1, h * TB + t);  // dP . v_n (v = xhat or x) + the additive term
          const float a = __builtin_amdgcn_exp2f((rl(e0, h * TB + t) - m[h]) * IQ_LOG2E) * l[h];
          const float dS = (n0 + t < N) ? a * (dA - delta[h]) : 0.f;
          outv = lane == h * TB + t ? dS : outv;
          const float wt = dS * rstd[t];
          cacc[h] = fmaf(wt, mean[t], cacc[h]);
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[h][j] += wt * xv[t][j];
        }
      }
    }
    float* dst = BWD ? p.dSout : p.S;
    if (dst && elane && (n0 + et) < N) dst[erow + n0 + et] = outv;
  };
  // two register sets: the loads of the next iteration are in flight while this one is computed
  {
    constexpr int STEP = IQF_NW * TB;
    f4 xa[TB][CPL], xc[TB][CPL];
    float ta = 0.f, tc = 0.f, a0 = 0.f, a1 = 0.f, c0 = 0.f, c1 = 0.f;
    int n0 = w * TB;
    if (n0 < N) load(n0, xa, ta, a0, a1);
    while (n0 < N) {
      const int n1 = n0 + STEP;
      if (n1 < N) load(n1, xc, tc, c0, c1);
      compute(n0, xa, ta, a0, a1);
      if (n1 >= N) break;
      const int n2 = n1 + STEP;
      if (n2 < N) load(n2, xa, ta, a0, a1);
      compute(n1, xc, tc, c0, c1);
      n0 = n2;
    }
  }

  // ---- merge the waves head by head (wave h % NW ends up with head h) and store ----
#pragma unroll
  for (int h = 0; h < HQ; ++h) {
    if (h >= Q) continue;
    float* rec = rec_lds + ((size_t)w * 64 + lane) * REC;
    __syncthreads();                                   // the previous head's records have been consumed
    rec[0] = m[h]; rec[1] = l[h]; rec[2] = cacc[h];
#pragma unroll
    for (int j = 0; j < CPL; ++j) *reinterpret_cast<f4*>(rec + 4 + 4 * j) = acc[h][j];
    __syncthreads();
    if (w != (h % IQF_NW)) continue;
    float mm = -INFINITY, ll = 0.f, cc = 0.f;
    f4 aa[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) aa[j] = z4;
    for (int i = 0; i < IQF_NW; ++i) {                 // fixed order 0..NW-1 whichever wave merges
      const float* o = rec_lds + ((size_t)i * 64 + lane) * REC;
      if constexpr (!BWD) {
        const float mi = o[0];
        if (mi > -INFINITY) {
          const float mn = fmaxf(mm, mi);
          const float f0 = __builtin_amdgcn_exp2f((mm - mn) * IQ_LOG2E), fi = __builtin_amdgcn_exp2f((mi - mn) * IQ_LOG2E);
          ll = ll * f0 + o[1] * fi;
          cc = cc * f0 + o[2] * fi;
#pragma unroll
          for (int j = 0; j < CPL; ++j) aa[j] = aa[j] * f0 + *reinterpret_cast<const f4*>(o + 4 + 4 * j) * fi;
          mm = mn;
        }
      } else {
        cc += o[2];
#pragma unroll
        for (int j = 0; j < CPL; ++j) aa[j] += *reinterpret_cast<const f4*>(o + 4 + 4 * j);
      }
    }
    if constexpr (!BWD) {
      const float inv = 1.0f / ll;
      float* Pq = p.P + ((int64_t)b * Q + h) * D;
#pragma unroll
      for (int j = 0; j < CPL; ++j)
        if (cv[j]) *reinterpret_cast<f4*>(Pq + ch[j]) = (aa[j] - cc) * inv;
      if (lane == 0) *reinterpret_cast<f4*>(p.ML + ((int64_t)b * Q + h) * 4) = f4{mm, ll, 0.f, 0.f};
    } else {
      float* gq = dq + ((int64_t)b * Q + h) * D;
#pragma unroll
      for (int j = 0; j < CPL; ++j)
        if (cv[j]) *reinterpret_cast<f4*>(gq + ch[j]) = (aa[j] - cc) * p.scale;
    }
  }
}

static int iqf_cpl(int D) { return (D / 4 + 63) / 64; }
bool imgqf_supported(const PoolParams& p) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_IMGQF"); on = e ? atoi(e) : 1; }
  return on && p.cls_bstride != 0 && p.Q >= 1 && p.Q <= 4 && p.D % 4 == 0 && iqf_cpl(p.D) <= 5 && p.N >= 1;
}

template <int CPL, int TB>
static int iqf_launch(const PoolParams& p, float* dq, bool bwd, hipStream_t st) {
  const dim3 grid(p.B), block(IQF_NW * 64);
  if (p.x_bf16) {
    if (bwd) hipLaunchKernelGGL((ep_imgqf_kernel<CPL, 4, TB, true, true>), grid, block, 0, st, p, dq);
    else hipLaunchKernelGGL((ep_imgqf_kernel<CPL, 4, TB, true, false>), grid, block, 0, st, p, dq);
  } else {
    if (bwd) hipLaunchKernelGGL((ep_imgqf_kernel<CPL, 4, TB, false, true>), grid, block, 0, st, p, dq);
    else hipLaunchKernelGGL((ep_imgqf_kernel<CPL, 4, TB, false, false>), grid, block, 0, st, p, dq);
  }
  EP_LAUNCH_CHECK(bwd ? "ep_imgqf_kernel (backward)" : "ep_imgqf_kernel (forward)");
  return 0;
}

static int iqf_dispatch(const PoolParams& p, float* dq, bool bwd, hipStream_t st) {
  EP_REQUIRE(imgqf_supported(p), EP_E_UNSUPPORTED, "full-width per-image-query token pass: Q=%d D=%d not supported", p.Q, p.D);
  switch (iqf_cpl(p.D)) {
    case 1: return iqf_launch<1, 4>(p, dq, bwd, st);
    case 2: return iqf_launch<2, 4>(p, dq, bwd, st);
    case 3: return iqf_launch<3, 4>(p, dq, bwd, st);
    case 4: return iqf_launch<4, 2>(p, dq, bwd, st);
    default: return iqf_launch<5, 2>(p, dq, bwd, st);
  }
}
int imgqf_forward(const PoolParams& p, hipStream_t st) { return iqf_dispatch(p, nullptr, false, st); }
int imgqf_backward(const PoolParams& p, float* dq, hipStream_t st) { return iqf_dispatch(p, dq, true, st); }

}  // namespace ep
