// EP token passes for WIDE rows (D = 2048 or 4096: ViT-g / DINOv3 ViT-7B tokens, BASELINE configs[4]).
//
// At D = 4096 one token row is 16 KiB and the pooled state of an image is Q x D = 128 KiB, so the
// "every wave owns full rows" layout of ep_pool_stream.hip no longer fits in registers or LDS.  Here the
// row is split ACROSS the 8 waves of a workgroup: wave w owns the D/8-float slice [w*Ds, (w+1)*Ds) of every
// row, for ALL Q <= 8 queries:
//   * each wave streams only its own slice HBM -> LDS with LDS-DMA into a PRIVATE ring (a per-lane FIFO:
//     every lane reads back exactly the 16 bytes it fetched, so there is no swizzle, no bank conflict and
//     no workgroup barrier for the data path at all); completion is tracked with a counted vmcnt;
//   * per mini-batch of 2 tokens a wave computes its slice's partial scores for the 16 (query, token)
//     pairs, reduce-scatters them over its 64 lanes (permlane32/16 swaps + DPP), and the 8 waves exchange
//     16 floats each through a double-buffered 1 KiB LDS scratch -- ONE s_barrier per 32 KiB of tokens;
//   * every wave then holds all 16 scores (lane = pair), runs the lazy-max online softmax redundantly, and
//     pools its own slice: acc[q][slice] += a[q][t] * x[t][slice] with the weights broadcast to SGPRs.
// The backward pass has the same shape (dA partials instead of scores, gradient accumulators instead of
// the pooled state); its S / ML inputs ride in the ring as one extra 4-byte-per-lane DMA per item.
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"

namespace ep {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// waves per workgroup = slices per row: 8 (D = 2048, 4096) or 4 (D = 1024)
constexpr int WQ = 8;                  // queries handled (Q <= 8)
constexpr int WTB = 2;                 // tokens per ring item / mini-batch
constexpr float W_LOG2E = 1.4426950408889634f;
constexpr float W_LAZY = 12.0f;

// BF16: tokens stored as bf16.  A lane then owns 8 CONSECUTIVE elements of the slice per 16-byte DMA piece (two
// 4-element register chunks), so the fp32 side arrays (queries, pooled state, dP, gradient partials) are addressed
// with the matching element map `eoff`; KPW must be even.
template <int KPW, int WNW, bool BF16> struct WideCfg {
  static constexpr int DPT = BF16 ? KPW / 2 : KPW;                        // 1 KiB DMA pieces per token per wave
  static constexpr int ITEM_TOK_BYTES = WTB * DPT * 1024;                 // this wave's slice of two token rows
  static constexpr int ITEM_BYTES = ITEM_TOK_BYTES + 256;                 // + one 4-byte-per-lane piece (backward)
  static constexpr int NITEM = (DPT == 1) ? 8 : 4;                         // ring depth per wave
  static constexpr int WG_PER_CU = (WNW == 4) ? 2 : 1;
  static constexpr int KD_F = WTB * DPT;                                  // DMA instructions per item, forward
  static constexpr int KD_B = WTB * DPT + 1;
  static_assert(!BF16 || KPW % 2 == 0, "bf16 wide rows: KPW must be even");
  // element offset (inside the wave's slice) of register chunk k of lane `lane`
  static __device__ __forceinline__ int eoff(int k, int lane) {
    return BF16 ? (k >> 1) * 512 + lane * 8 + (k & 1) * 4 : k * 256 + lane * 4;
  }
  // read this lane's data of token t of an item into KPW fp32 chunks
  static __device__ __forceinline__ void read_tok(const char* item, int t, int lane, f4 (&x)[KPW]) {
    if (BF16) {
#pragma unroll
      for (int j = 0; j < KPW / 2; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(item + (t * DPT + j) * 1024 + lane * 16);
        x[2 * j] = bf16x4_to_f4(uint2{v.x, v.y});
        x[2 * j + 1] = bf16x4_to_f4(uint2{v.z, v.w});
      }
    } else {
#pragma unroll
      for (int k = 0; k < KPW; ++k) x[k] = *reinterpret_cast<const f4*>(item + (t * KPW + k) * 1024 + lane * 16);
    }
  }
  // issue the DMA pieces of token row `row` (element pointer semantics: byte address of the slice start)
  static __device__ __forceinline__ void dma_tok(const char* slice_row, char* slot, int t, int lane) {
#pragma unroll
    for (int j = 0; j < DPT; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(slice_row + j * 1024 + lane * 16), (lds_ptr_t)(slot + (t * DPT + j) * 1024), 16, 0,
                                       EP_DMA_AUX);
  }
  static constexpr size_t LDS_BYTES = (size_t)WNW * NITEM * ITEM_BYTES + 2 * WNW * WQ * WTB * 4;
};

template <int N>
__device__ __forceinline__ void wwait_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wwait(int n) {
#define EP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    EP_W(0) EP_W(1) EP_W(2) EP_W(3) EP_W(4) EP_W(5) EP_W(6) EP_W(7) EP_W(8) EP_W(9) EP_W(10) EP_W(11) EP_W(12)
    EP_W(13) EP_W(14) EP_W(15) EP_W(16) EP_W(17) EP_W(18) EP_W(19) EP_W(20) EP_W(21) EP_W(22) EP_W(23) EP_W(24)
    EP_W(25) EP_W(26) EP_W(27) EP_W(28) EP_W(29) EP_W(30) EP_W(31) EP_W(32) EP_W(33) EP_W(34) EP_W(35)
    default: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
  }
#undef EP_W
}
__device__ __forceinline__ void wbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f2 wfma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// 16 lane-partials pin[v], v = 2*q + t  ->  in every lane, the 8-wave total of pair v = lane & 15.
// Stage 1 (inside the wave): fold across 32-lane halves, 16-lane rows, then inside rows; register i of the
// folded set holds, in row r, the wave-sum of pin[i + 4*(r&1) + 8*(r>>1)].  Stage 2: 16 floats per wave
// through LDS, one workgroup barrier, every wave adds the 8 contributions.
template <int WNW>
__device__ __forceinline__ float reduce_pairs(const float (&pin)[WQ * WTB], float* scratch_buf, int w, int lane) {
  float r8[8], r4[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) r8[i] = fold32(pin[i], pin[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) r4[i] = row16_sum(fold16(r8[i], r8[i + 4]));
  const int li = lane & 15, row = lane >> 4;
  const float mine = li == 0 ? r4[0] : (li == 1 ? r4[1] : (li == 2 ? r4[2] : r4[3]));
  if (li < 4) scratch_buf[w * 16 + li + 4 * (row & 1) + 8 * (row >> 1)] = mine;
  wbarrier();
  float s = scratch_buf[row * 16 + li];                                          // wave `row` (+ wave row+4)
  if (WNW == 8) s += scratch_buf[(row + 4) * 16 + li];
  s = fold16(s, s);
  s = fold32(s, s);
  return s;
}

template <int KPW>
__device__ __forceinline__ void wide_partials(const f4 (&w)[WQ][KPW], const f4 (&xv)[WTB][KPW], float (&pin)[WQ * WTB]) {
#pragma unroll
  for (int q = 0; q < WQ; ++q)
#pragma unroll
    for (int t = 0; t < WTB; ++t) {
      f2 s = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KPW; ++k) {
        s = wfma2(w[q][k].xy, xv[t][k].xy, s);
        s = wfma2(w[q][k].zw, xv[t][k].zw, s);
      }
      pin[2 * q + t] = s.x + s.y;
    }
}

// ---------------------------------------------------------------------------------------
template <int KPW, int WNW, bool BF16>
__global__ __launch_bounds__(WNW * 64, 2) void ep_pool_wide_fwd_kernel(PoolParams p) {
  using Cfg = WideCfg<KPW, WNW, BF16>;
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int NITEM = Cfg::NITEM, KD = Cfg::KD_F;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const int Ds = KPW * 256;
  const int sbase = w * Ds;                                   // first element of this wave's slice inside a row
  char* ring = lds + (size_t)w * NITEM * Cfg::ITEM_BYTES;
  float* scratch = reinterpret_cast<float*>(lds + (size_t)WNW * NITEM * Cfg::ITEM_BYTES);
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int items_per_img = (N + WTB - 1) / WTB;
  const int n_items = n_img * items_per_img;
  if (n_items <= 0) return;

  f4 cq[WQ][KPW];
  auto load_cls = [&](int b) {
#pragma unroll
    for (int q = 0; q < WQ; ++q)
#pragma unroll
      for (int k = 0; k < KPW; ++k) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < Q) v = *reinterpret_cast<const f4*>(p.cls + (int64_t)b * p.cls_bstride + (int64_t)q * D + sbase + Cfg::eoff(k, lane));
        cq[q][k] = v * p.scale;
      }
  };
  load_cls(wg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, pit = 0, pslot = 0;
  auto produce = [&]() {
    if (pi < n_items) {
      const int b = wg + pimg * G;
      const char* src = reinterpret_cast<const char*>(p.x) + (EP_IMG_OFF(p, b) + sbase) * ES;
      char* slot = ring + pslot * Cfg::ITEM_BYTES;         // wave-uniform; lane l's 16 bytes land at +16*l
#pragma unroll
      for (int t = 0; t < WTB; ++t) {
        int n = pit * WTB + t; n = n < N ? n : N - 1;          // odd N: the surplus row re-reads the last token
        Cfg::dma_tok(src + (int64_t)n * D * ES, slot, t, lane);
      }
      ++pi;
      pslot = (pslot + 1 == NITEM) ? 0 : pslot + 1;
      if (++pit == items_per_img) { pit = 0; ++pimg; }
    }
  };
#pragma unroll
  for (int s = 0; s < NITEM; ++s) produce();

  f4 acc[WQ][KPW];
  float m = -INFINITY, mL = -INFINITY, lsum = 0.f;            // online-softmax state of this lane's query (lane&15)>>1
  const int myq = (lane & 15) >> 1, myt = lane & 1;
  int cimg = 0, cit = 0, cslot = 0;
  for (int i = 0; i < n_items; ++i) {
    const int ahead = pi - 1 - i;                             // items issued after item i
    if (ahead == NITEM - 1) wwait_imm<(NITEM - 1) * KD>(); else wwait(ahead * KD);
    const int b = wg + cimg * G;
    const int n0 = cit * WTB;
    const int nvalid = (N - n0) < WTB ? (N - n0) : WTB;
    if (cit == 0) {
      if (p.cls_bstride != 0 && cimg != 0) { load_cls(b); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      m = -INFINITY; mL = -INFINITY; lsum = 0.f;
#pragma unroll
      for (int q = 0; q < WQ; ++q)
#pragma unroll
        for (int k = 0; k < KPW; ++k) acc[q][k] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const char* item = ring + cslot * Cfg::ITEM_BYTES;
    f4 xv[WTB][KPW];
#pragma unroll
    for (int t = 0; t < WTB; ++t) Cfg::read_tok(item, t, lane, xv[t]);
    float pin[WQ * WTB];
    wide_partials<KPW>(cq, xv, pin);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot's data is in registers: refill it
    cslot = (cslot + 1 == NITEM) ? 0 : cslot + 1;
    produce();
    const float u = reduce_pairs<WNW>(pin, scratch + (i & 1) * (WNW * 16), w, lane);     // score of pair (myq, myt)
    const bool valid = myt < nvalid;
    const float ue = valid ? u : -INFINITY;
    const float um = fmaxf(ue, dpp_f<0xB1>(ue));              // max over the two tokens of my query
    if (__builtin_amdgcn_ballot_w64(um > m + W_LAZY) != 0ull) {   // wave-uniform, rare: move the running max
      const float mn = fmaxf(m, um);
      const float f = __builtin_amdgcn_exp2f((m - mn) * W_LOG2E);
      m = mn; mL = mn * W_LOG2E; lsum *= f;
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const float fq = readlane_f(f, 2 * q);
#pragma unroll
        for (int k = 0; k < KPW; ++k) acc[q][k] *= fq;
      }
    }
    const float pr = __builtin_amdgcn_exp2f(fmaf(ue, W_LOG2E, -mL));
    lsum += pr;
    if (w == 0 && lane < 16 && valid && myq < Q) p.S[((int64_t)b * QS + myq) * N + n0 + myt] = u;
#pragma unroll
    for (int q = 0; q < WQ; ++q)
#pragma unroll
      for (int t = 0; t < WTB; ++t) {
        const float a = readlane_f(pr, 2 * q + t);            // 0 for a padded token
#pragma unroll
        for (int k = 0; k < KPW; ++k) acc[q][k] += a * xv[t][k];
      }
    if (cit == items_per_img - 1) {
      const float l = lsum + dpp_f<0xB1>(lsum);
      const float inv = 1.0f / l;
#pragma unroll
      for (int q = 0; q < WQ; ++q)
        if (q < Q) {
          const float iq = readlane_f(inv, 2 * q);
          float* Pq = p.P + ((int64_t)b * QS + q) * D + sbase;
#pragma unroll
          for (int k = 0; k < KPW; ++k) *reinterpret_cast<f4*>(Pq + Cfg::eoff(k, lane)) = acc[q][k] * iq;
        }
      if (w == 0 && lane < 16 && myt == 0 && myq < Q) {
        const f4 rec = {m, l, 0.f, 0.f};
        *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + myq) * 4) = rec;
      }
      cit = 0; ++cimg;
    } else {
      ++cit;
    }
  }
}

// ---------------------------------------------------------------------------------------
template <int KPW, int WNW, bool BF16>
__global__ __launch_bounds__(WNW * 64, 2) void ep_pool_wide_bwd_kernel(PoolParams p) {
  using Cfg = WideCfg<KPW, WNW, BF16>;
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int NITEM = Cfg::NITEM, KD = Cfg::KD_B;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const int Ds = KPW * 256;
  const int sbase = w * Ds;
  char* ring = lds + (size_t)w * NITEM * Cfg::ITEM_BYTES;
  float* scratch = reinterpret_cast<float*>(lds + (size_t)WNW * NITEM * Cfg::ITEM_BYTES);
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int items_per_img = (N + WTB - 1) / WTB;
  const int n_items = n_img * items_per_img;
  const int myq = (lane & 15) >> 1, myt = lane & 1;
  const int sq = myq < Q ? myq : Q - 1;

  f4 gacc[WQ][KPW];
#pragma unroll
  for (int q = 0; q < WQ; ++q)
#pragma unroll
    for (int k = 0; k < KPW; ++k) gacc[q][k] = f4{0.f, 0.f, 0.f, 0.f};

  if (n_items > 0) {
    int pi = 0, pimg = 0, pit = 0, pslot = 0;
    auto produce = [&]() {
      if (pi < n_items) {
        const int b = wg + pimg * G;
        const char* src = reinterpret_cast<const char*>(p.x) + (EP_IMG_OFF(p, b) + sbase) * ES;
        char* slot = ring + pslot * Cfg::ITEM_BYTES;
#pragma unroll
        for (int t = 0; t < WTB; ++t) {
          int n = pit * WTB + t; n = n < N ? n : N - 1;
          Cfg::dma_tok(src + (int64_t)n * D * ES, slot, t, lane);
        }
        int nn = pit * WTB + myt; nn = nn < N ? nn : N - 1;   // raw score of my (query, token) pair
        __builtin_amdgcn_global_load_lds((gptr_t)(p.S + ((int64_t)b * QS + sq) * N + nn),
                                         (lds_ptr_t)(slot + Cfg::ITEM_TOK_BYTES), 4, 0, 0);
        ++pi;
        pslot = (pslot + 1 == NITEM) ? 0 : pslot + 1;
        if (++pit == items_per_img) { pit = 0; ++pimg; }
      }
    };
#pragma unroll
    for (int s = 0; s < NITEM; ++s) produce();

    f4 gq[WQ][KPW];
    float mLq = 0.f, il = 0.f, dl = 0.f;
    int cimg = 0, cit = 0, cslot = 0;
    for (int i = 0; i < n_items; ++i) {
      const int b = wg + cimg * G;
      if (cit == 0) {
        // image header: this wave's slice of dP[b] and the softmax statistics of my query (plain loads; the
        // counted wait below then also covers them -- one shallow bubble per image, ~1 % at 196 x 4096)
#pragma unroll
        for (int q = 0; q < WQ; ++q)
#pragma unroll
          for (int k = 0; k < KPW; ++k) {
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < Q) v = *reinterpret_cast<const f4*>(p.dP + ((int64_t)b * QS + q) * D + sbase + Cfg::eoff(k, lane));
            gq[q][k] = v;
          }
        const f4 ml = *reinterpret_cast<const f4*>(p.ML + ((int64_t)b * QS + sq) * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mLq = ml.x * W_LOG2E; il = 1.0f / ml.y; dl = ml.z;
      } else {
        const int ahead = pi - 1 - i;
        if (ahead == NITEM - 1) wwait_imm<(NITEM - 1) * KD>(); else wwait(ahead * KD);
      }
      const int n0 = cit * WTB;
      const int nvalid = (N - n0) < WTB ? (N - n0) : WTB;
      const char* item = ring + cslot * Cfg::ITEM_BYTES;
      f4 xv[WTB][KPW];
#pragma unroll
      for (int t = 0; t < WTB; ++t) Cfg::read_tok(item, t, lane, xv[t]);
      const float sraw = *reinterpret_cast<const float*>(item + Cfg::ITEM_TOK_BYTES + lane * 4);
      float pin[WQ * WTB];
      wide_partials<KPW>(gq, xv, pin);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      cslot = (cslot + 1 == NITEM) ? 0 : cslot + 1;
      produce();
      const float dA = reduce_pairs<WNW>(pin, scratch + (i & 1) * (WNW * 16), w, lane);
      const float a = __builtin_amdgcn_exp2f(fmaf(sraw, W_LOG2E, -mLq)) * il;
      const float wgt = (myt < nvalid && myq < Q) ? a * (dA - dl) : 0.f;
#pragma unroll
      for (int q = 0; q < WQ; ++q)
#pragma unroll
        for (int t = 0; t < WTB; ++t) {
          const float g = readlane_f(wgt, 2 * q + t);
#pragma unroll
          for (int k = 0; k < KPW; ++k) gacc[q][k] += g * xv[t][k];
        }
      if (++cit == items_per_img) { cit = 0; ++cimg; }
    }
  }
#pragma unroll
  for (int q = 0; q < WQ; ++q)
    if (q < Q) {
      float* Gq = p.Gpart + ((int64_t)wg * Q + q) * D + sbase;
#pragma unroll
      for (int k = 0; k < KPW; ++k) *reinterpret_cast<f4*>(Gq + Cfg::eoff(k, lane)) = gacc[q][k];
    }
}

// ---------------------------------------------------------------------------------------
bool wide_supported(int D, int Q, int64_t cls_bstride, int x_bf16) {
  (void)cls_bstride;
  static int d1024 = -1;
  if (d1024 < 0) { const char* e = getenv("EP_POOL_WIDE_1024"); d1024 = e ? atoi(e) : 0; }
  return (D == 2048 || D == 4096 || (D == 1024 && d1024 && !x_bf16)) && Q >= 1 && Q <= WQ;
}
int wide_grid(int D, int B, int x_bf16) {
  int g = cu_count() * ((D == 1024 || (D == 2048 && x_bf16)) ? 2 : 1);        // 4-wave variants: two workgroups per CU
  if (const char* e = getenv("EP_POOL_GRID")) { int v = atoi(e); if (v >= 1) g = v; }
  return g < B ? g : B;
}

template <int KPW, int WNW, bool BF16>
static int wide_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  using Cfg = WideCfg<KPW, WNW, BF16>;
  auto kf = ep_pool_wide_fwd_kernel<KPW, WNW, BF16>;
  auto kb = ep_pool_wide_bwd_kernel<KPW, WNW, BF16>;
  const void* fn = bwd ? (const void*)kb : (const void*)kf;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", Cfg::LDS_BYTES, hipGetErrorString(e)); return (int)e; }
  if (bwd) hipLaunchKernelGGL(kb, dim3(grid), dim3(WNW * 64), Cfg::LDS_BYTES, st, p);
  else hipLaunchKernelGGL(kf, dim3(grid), dim3(WNW * 64), Cfg::LDS_BYTES, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_wide_bwd_kernel" : "ep_pool_wide_fwd_kernel");
  return 0;
}

int wide_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  if (wideb_supported(p.D, p.Q, p.cls_bstride, p.x_bf16, bwd) && !p.tokstat) return wideb_launch(bwd, p, grid, st);
  if (p.x_bf16) {
    if (p.D == 2048) return wide_launch_one<2, 4, true>(bwd, p, grid, st);      // 4 waves x 512-element slices
    if (p.D == 4096) return wide_launch_one<2, 8, true>(bwd, p, grid, st);
  } else {
    if (p.D == 1024) return wide_launch_one<1, 4, false>(bwd, p, grid, st);
    if (p.D == 2048) return wide_launch_one<1, 8, false>(bwd, p, grid, st);
    if (p.D == 4096) return wide_launch_one<2, 8, false>(bwd, p, grid, st);
  }
  set_error("no wide-row kernel for D=%d", p.D);
  return EP_E_UNSUPPORTED;
}

}  // namespace ep
