// EP token passes for WIDE rows stored as bf16 (D = 4096: the pre-dumped DINOv3 ViT-7B tokens, BASELINE configs[4] in the only form
// that fits 8 x 288 GB) -- the HYBRID form: scores on the bf16 matrix cores, pooling on the vector ALU.  Round 5.
//
// ep_pool_wide.hip runs these rows on the vector ALU alone: ~125 instructions per token and wave, 530 - 590 us per pass =
// 0.36 - 0.38 of the bf16 bytes -- issue-bound (the same arithmetic on fp32 rows is at 0.66 of twice the bytes).  Of those
// instructions 32 are the score multiply-adds and ~25 the 16-pair reduce-scatter + exchange every TWO tokens.  Here, per
// 8-token tile, a wave (D-slice of 512 columns, as before) multiplies its slice of the queries -- pre-split once into three
// bf16 terms, A operand rows (q, hi) | (q, mid) and (q, lo) | 0 -- against the tile on v_mfma_f32_16x16x32_bf16: 32 matrix
// instructions give the wave's partial scores of all 8 x 8 (query, token) pairs at fp32 accuracy (tokens are exact in bf16;
// every fp32 query value is the exact sum of its three terms), the 8 waves exchange 256 bytes each through LDS (one barrier per
// 8 tokens instead of one per 2), every wave runs the online softmax on lane = pair and pools its own slice on the vector ALU
// with the weights broadcast through SGPRs (reference poolings/ep.py:41-44; the pooled state of an image is Q x D fp32 = 128 KiB:
// it stays in the waves' registers, 64 per lane).
//
// LDS: the token tile must be readable BOTH ways -- as the MFMA's B operand (lane (token, k-group) reads 16 bytes of ANOTHER
// lane's copy) and as the pooling operand (lane L reads its own 8 columns).  Each wave keeps a private ring of 8-token tiles,
// token-major rows of 1 KiB (its 512 columns), 16-byte chunk c of token t stored in slot c ^ t: the swizzle is applied on the
// SOURCE address of the LDS-DMA (lane L fetches chunk L ^ t), both reads are conflict-free, and because a wave only ever reads
// what it fetched itself the data path needs no workgroup barrier -- a counted vmcnt as in ep_pool_wide.hip.
#include "ep_planes_dev.h"
#include "ep_pool_stream.h"

namespace ep {

typedef __attribute__((address_space(3))) void* wb_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* wb_gptr_t;

constexpr int WB_TT = 8;                             // tokens per tile (half of the MFMA's 16 columns: a 16-token tile of all 8 slices is 128 KiB)
constexpr int WB_NW = 8;                             // waves = D-slices of 512 columns
constexpr int WB_KS = 16;                            // K-steps of 32 columns per slice
constexpr int WB_ROWB = 1024;                        // bytes of one token's slice
constexpr int WB_TILE_W = WB_TT * WB_ROWB;           // 8 KiB per wave and tile
constexpr int WB_SLOTB = WB_NW * WB_TILE_W;          // 64 KiB per ring slot (all waves)
constexpr int WB_NSLOT = 2;
constexpr int WB_SCR = WB_NW * 256;                  // one score-exchange buffer: 16 entries x 16 bytes per wave
constexpr size_t WB_LDS = (size_t)WB_NSLOT * WB_SLOTB + 2 * WB_SCR;     // 132 KiB
constexpr float WB_LOG2E = 1.4426950408889634f;
constexpr float WB_LAZY = 12.0f;
#ifndef EP_WB_ABLATE
#define EP_WB_ABLATE 0                               // diagnostic builds only: 1 no pooling, 2 no score MFMAs, 4 no exchange barrier / gather
#endif

__device__ __forceinline__ void wb_wait_tile(bool next_in_flight) {
  if (next_in_flight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// max / sum over the 8 lanes that share a query (lanes 8 q .. 8 q + 7), result in all of them
__device__ __forceinline__ float wb_max8(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v));
  return v;
}
__device__ __forceinline__ float wb_sum8(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v);
  return v;
}

// The three bf16 terms of this wave's query slice as MFMA A operands: lane (row r = lane & 15, k-group kk) holds, for K-step
// ks, the 8 columns 32 ks + 8 kk .. + 7 of row r: rows 0..7 = (query r, hi), rows 8..15 = (query r - 8, mid) in `hm`;
// rows 0..7 = (query r, lo), rows 8..15 = 0 in `lo`.  `src` = the fp32 rows (row stride ld), scaled by `scale`.
// `one` (PoolParams.nterms == 1, the AMP-bf16 arithmetic mode): the operand rounded to bf16 only -- rows 8..15 and `lo` are zero.
__device__ __forceinline__ void wb_query_terms(const float* src, int64_t ld, int nrows, float scale, int sbase, int lane,
                                               pl_u4 (&hm)[WB_KS], pl_u4 (&lo)[WB_KS], bool one) {
  const int r = lane & 15, kk = lane >> 4, q = r & 7;
#pragma unroll
  for (int ks = 0; ks < WB_KS; ++ks) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (q < nrows) {
      const float* s = src + (int64_t)q * ld + sbase + 32 * ks + 8 * kk;
      const f4 a = *reinterpret_cast<const f4*>(s), b = *reinterpret_cast<const f4*>(s + 4);
      v[0] = a.x * scale; v[1] = a.y * scale; v[2] = a.z * scale; v[3] = a.w * scale;
      v[4] = b.x * scale; v[5] = b.y * scale; v[6] = b.z * scale; v[7] = b.w * scale;
    }
    pl_u4 t[3];
    pl_split8(v, t);
    hm[ks] = r < 8 ? t[0] : (one ? pl_u4{0u, 0u, 0u, 0u} : t[1]);
    lo[ks] = (r < 8 && !one) ? t[2] : pl_u4{0u, 0u, 0u, 0u};
  }
}

// partial scores of this wave's slice for the 8 x 8 (query, token) pairs of the tile -> its 16 entries of the exchange buffer
// (entry 8 h + t = queries 4 h .. 4 h + 3 of token t)
__device__ __forceinline__ void wb_scores(const char* tile, const pl_u4 (&hm)[WB_KS], const pl_u4 (&lo)[WB_KS], char* scr_w, int lane, bool one) {
  const int t = lane & 7, kk = lane >> 4;            // (lanes 8..15 of a row repeat tokens 0..7: the MFMA has 16 columns)
  // One accumulator chain, the B operand of K-step ks + 1 fetched before the two matrix instructions of K-step ks (measured on
  // one box, 1024 x 196 x 4096: read -> wait -> multiply per K-step 455 us; two chains with two operands in flight 504 us and
  // four chains 18 spilled registers -- the 128 registers of query terms and the 64 of pooled state leave ~60 for everything else)
  // chunk 4 ks + kk of token t sits in slot (4 ks + kk) ^ t = 4 (ks ^ (t >> 2)) + (kk ^ (t & 3)): two lane bases (even / odd K-steps)
  // + an immediate per K-step instead of sixteen address registers
  const int tb = (t >> 2) << 6;
  const char* be = tile + t * WB_ROWB + ((kk ^ (t & 3)) << 4) + tb;
  const char* bo = be - 2 * tb;
  f4v tot = {0.f, 0.f, 0.f, 0.f};
  pl_u4 b0 = *reinterpret_cast<const pl_u4*>(be), b1;
#pragma unroll
  for (int ks = 0; ks < WB_KS; ks += 2) {
    b1 = *reinterpret_cast<const pl_u4*>(bo + 64 * (ks + 1));
    if constexpr (!(EP_WB_ABLATE & 2)) { if (!one) tot = pl_mfma(lo[ks], b0, tot); tot = pl_mfma(hm[ks], b0, tot); }
    else tot[0] += __uint_as_float(b0[0] ^ lo[ks][1] ^ hm[ks][2]);
    if (ks + 2 < WB_KS) b0 = *reinterpret_cast<const pl_u4*>(be + 64 * (ks + 2));
    if constexpr (!(EP_WB_ABLATE & 2)) { if (!one) tot = pl_mfma(lo[ks + 1], b1, tot); tot = pl_mfma(hm[ks + 1], b1, tot); }
    else tot[1] += __uint_as_float(b1[0] ^ lo[ks + 1][1] ^ hm[ks + 1][2]);
  }
  // rows 0..7 (k-groups 0, 1) hold hi + lo, rows 8..15 (k-groups 2, 3) mid: add the two 32-lane halves
  f4v s;
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = fold32(tot[r], tot[r]);
  if (lane < 32 && (lane & 15) < 8) *reinterpret_cast<f4v*>(scr_w + ((kk << 3) + t) * 16) = s;
}
// total over the 8 waves of pair (q, t) = (lane >> 3, lane & 7)
__device__ __forceinline__ float wb_gather(const char* scr, int lane) {
  const int q = lane >> 3, t = lane & 7;
  const char* e = scr + (((q >> 2) << 3) + t) * 16 + (q & 3) * 4;
  float u = 0.f;
#pragma unroll
  for (int w = 0; w < WB_NW; ++w) u += *reinterpret_cast<const float*>(e + w * 256);
  return u;
}
// this lane's 8 columns of token t of the tile, widened to fp32
__device__ __forceinline__ void wb_read_tok(const char* tile, int t, int lane, f4& x0, f4& x1) {
  const uint4 v = *reinterpret_cast<const uint4*>(tile + t * WB_ROWB + ((lane ^ t) << 4));
  x0 = bf16x4_to_f4(uint2{v.x, v.y});
  x1 = bf16x4_to_f4(uint2{v.z, v.w});
}

__global__ __launch_bounds__(WB_NW * 64, 1) void ep_pool_wideb_fwd_kernel(PoolParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;
  const int sbase = w * (32 * WB_KS);
  char* ring = lds + w * WB_TILE_W;
  char* scr = lds + WB_NSLOT * WB_SLOTB;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int tiles_per_img = (N + WB_TT - 1) / WB_TT;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;

  pl_u4 qhm[WB_KS], qlo[WB_KS];
  const bool one = p.nterms == 1;
  wb_query_terms(p.cls, D, Q, p.scale, sbase, lane, qhm, qlo, one);

  int pi = 0, pimg = 0, ptile = 0;
  auto produce = [&]() {
    if (pi < n_items) {
      const int b = wg + pimg * G;
      const char* src = reinterpret_cast<const char*>(p.x) + (EP_IMG_OFF(p, b) + sbase) * 2;
      char* slot = ring + (pi & 1) * WB_SLOTB;
#pragma unroll
      for (int t = 0; t < WB_TT; ++t) {
        int n = ptile * WB_TT + t; n = n < N ? n : N - 1;      // the ragged last tile re-reads the last token (masked below)
        __builtin_amdgcn_global_load_lds((wb_gptr_t)(src + (int64_t)n * D * 2 + ((lane ^ t) << 4)), (wb_lds_ptr_t)(slot + t * WB_ROWB), 16, 0, EP_DMA_AUX);
      }
      ++pi;
      if (++ptile == tiles_per_img) { ptile = 0; ++pimg; }
    }
  };
  produce(); produce();

  f4 acc[8][2];                                      // pooled state: [query][this lane's 8 columns]
  float m = -INFINITY, mL = -INFINITY, lsum = 0.f;   // online softmax of pair lane (q, t) = (lane >> 3, lane & 7)
  const int myq = lane >> 3, myt = lane & 7;
  int cimg = 0, ctile = 0;
  for (int i = 0; i < n_items; ++i) {
    wb_wait_tile(i + 1 < pi);
    const int b = wg + cimg * G;
    const int n0 = ctile * WB_TT;
    const int nvalid = (N - n0) < WB_TT ? (N - n0) : WB_TT;
    if (ctile == 0) {
      m = -INFINITY; mL = -INFINITY; lsum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { acc[q][0] = f4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f4{0.f, 0.f, 0.f, 0.f}; }
    }
    const char* tile = ring + (i & 1) * WB_SLOTB;
    char* sb = scr + (i & 1) * WB_SCR;
    wb_scores(tile, qhm, qlo, sb + w * 256, lane, one);
    float u;
    if constexpr (EP_WB_ABLATE & 4) u = *reinterpret_cast<const float*>(sb + w * 256 + (lane & 15) * 16);
    else { wb_barrier(); u = wb_gather(sb, lane); }
    const bool valid = myt < nvalid;
    const float ue = valid ? u : -INFINITY;
    const float um = wb_max8(ue);
    if (__builtin_amdgcn_ballot_w64(um > m + WB_LAZY) != 0ull) {         // wave-uniform, rare: move the running max
      const float mn = fmaxf(m, um);
      const float f = __builtin_amdgcn_exp2f((m - mn) * WB_LOG2E);
      m = mn; mL = mn * WB_LOG2E; lsum *= f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float fq = readlane_f(f, 8 * q);
        acc[q][0] *= fq; acc[q][1] *= fq;
      }
    }
    const float pr = __builtin_amdgcn_exp2f(fmaf(ue, WB_LOG2E, -mL));   // 0 for a padded token
    lsum += pr;
    if (w == 0 && valid && myq < Q) p.S[((int64_t)b * QS + myq) * N + n0 + myt] = u;
#pragma unroll 2                                     // (fully unrolled the compiler keeps all eight tokens' columns live: 25 - 42 spilled registers)
    for (int t = 0; t < WB_TT; ++t) {
      f4 x0, x1;
      wb_read_tok(tile, t, lane, x0, x1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float a = readlane_f(pr, 8 * q + t);
        if constexpr (EP_WB_ABLATE & 1) { if (q == 0) { acc[q][0] += a * x0; acc[q][1] += a * x1; } }
        else { acc[q][0] += a * x0; acc[q][1] += a * x1; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // this wave is done with its slot: refill it
    if (ctile == tiles_per_img - 1) {
      const float l = wb_sum8(lsum);
      const float inv = 1.0f / l;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < Q) {
          const float iq = readlane_f(inv, 8 * q);
          float* Pq = p.P + ((int64_t)b * QS + q) * D + sbase + 8 * lane;
          *reinterpret_cast<f4*>(Pq) = acc[q][0] * iq;
          *reinterpret_cast<f4*>(Pq + 4) = acc[q][1] * iq;
        }
      if (w == 0 && myt == 0 && myq < Q) {
        const f4 rec = {m, l, 0.f, 0.f};
        *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + myq) * 4) = rec;
      }
      ctile = 0; ++cimg;
    } else {
      ++ctile;
    }
    produce();                                                           // tile i + 2 into the slot tile i leaves (the LAST vector-memory
                                                                         // operations of the iteration: wb_wait_tile counts on it)
  }
}

// Second pass (reference poolings/ep.py:41-44 under autograd; same shape as the forward): per image the wave's slice of dP[b] is
// split into the three-term A operands (what the queries are in the forward: +8 % instructions per image), dA = dP . x comes from the
// matrix cores, the softmax weights are rebuilt from the saved raw scores S and {max, sum, delta} of ML, and
// G[q] += a (dA - delta) x accumulates on the vector ALU over ALL images of the workgroup (its partial of dcls, reduced later).
// The raw scores of a tile ride in the ring as a ninth 4-byte-per-lane DMA (lane = pair).
constexpr int WB_SRAW = WB_NW * 256;                 // per slot: one 256-byte score record per wave
constexpr size_t WB_LDS_BWD = WB_LDS + 2 * WB_SRAW;

__global__ __launch_bounds__(WB_NW * 64, 1) void ep_pool_wideb_bwd_kernel(PoolParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;
  const int sbase = w * (32 * WB_KS);
  char* ring = lds + w * WB_TILE_W;
  char* scr = lds + WB_NSLOT * WB_SLOTB;
  char* sraw_ring = lds + WB_LDS + w * 256;            // slot s at + s * WB_SRAW
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int tiles_per_img = (N + WB_TT - 1) / WB_TT;
  const int n_items = n_img * tiles_per_img;
  const int myq = lane >> 3, myt = lane & 7;
  const int sq = myq < Q ? myq : Q - 1;
  const bool one = p.nterms == 1;

  f4 gacc[8][2];
#pragma unroll
  for (int q = 0; q < 8; ++q) { gacc[q][0] = f4{0.f, 0.f, 0.f, 0.f}; gacc[q][1] = f4{0.f, 0.f, 0.f, 0.f}; }

  if (n_items > 0) {
    int pi = 0, pimg = 0, ptile = 0;
    auto produce = [&]() {
      if (pi < n_items) {
        const int b = wg + pimg * G;
        const char* src = reinterpret_cast<const char*>(p.x) + (EP_IMG_OFF(p, b) + sbase) * 2;
        char* slot = ring + (pi & 1) * WB_SLOTB;
#pragma unroll
        for (int t = 0; t < WB_TT; ++t) {
          int n = ptile * WB_TT + t; n = n < N ? n : N - 1;
          __builtin_amdgcn_global_load_lds((wb_gptr_t)(src + (int64_t)n * D * 2 + ((lane ^ t) << 4)), (wb_lds_ptr_t)(slot + t * WB_ROWB), 16, 0, EP_DMA_AUX);
        }
        int nn = ptile * WB_TT + myt; nn = nn < N ? nn : N - 1;            // raw score of my (query, token) pair
        __builtin_amdgcn_global_load_lds((wb_gptr_t)(p.S + ((int64_t)b * QS + sq) * N + nn), (wb_lds_ptr_t)(sraw_ring + (pi & 1) * WB_SRAW), 4, 0, 0);
        ++pi;
        if (++ptile == tiles_per_img) { ptile = 0; ++pimg; }
      }
    };
    produce(); produce();

    pl_u4 ghm[WB_KS], glo[WB_KS];
    float mLq = 0.f, il = 0.f, dl = 0.f;
    int cimg = 0, ctile = 0;
    for (int i = 0; i < n_items; ++i) {
      const int b = wg + cimg * G;
      if (ctile == 0) {
        // image header: this wave's slice of dP[b] as matrix operands and the softmax statistics of my query (plain loads: the
        // wait for them also drains the DMA in flight -- one bubble per image)
        wb_query_terms(p.dP + (int64_t)b * QS * D, D, Q, 1.0f, sbase, lane, ghm, glo, one);
        const f4 ml = *reinterpret_cast<const f4*>(p.ML + ((int64_t)b * QS + sq) * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mLq = ml.x * WB_LOG2E; il = 1.0f / ml.y; dl = ml.z;
      } else {
        if (i + 1 < pi) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const int n0 = ctile * WB_TT;
      const int nvalid = (N - n0) < WB_TT ? (N - n0) : WB_TT;
      const char* tile = ring + (i & 1) * WB_SLOTB;
      char* sb = scr + (i & 1) * WB_SCR;
      const float sraw = *reinterpret_cast<const float*>(sraw_ring + (i & 1) * WB_SRAW + lane * 4);
      wb_scores(tile, ghm, glo, sb + w * 256, lane, one);
      wb_barrier();
      const float dA = wb_gather(sb, lane);
      const float a = __builtin_amdgcn_exp2f(fmaf(sraw, WB_LOG2E, -mLq)) * il;
      const float wgt = (myt < nvalid && myq < Q) ? a * (dA - dl) : 0.f;
#pragma unroll 2
      for (int t = 0; t < WB_TT; ++t) {
        f4 x0, x1;
        wb_read_tok(tile, t, lane, x0, x1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float g = readlane_f(wgt, 8 * q + t);
          gacc[q][0] += g * x0; gacc[q][1] += g * x1;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (++ctile == tiles_per_img) { ctile = 0; ++cimg; }
      produce();
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q < Q) {
      float* Gq = p.Gpart + ((int64_t)wg * Q + q) * D + sbase + 8 * lane;
      *reinterpret_cast<f4*>(Gq) = gacc[q][0];
      *reinterpret_cast<f4*>(Gq + 4) = gacc[q][1];
    }
}

bool wideb_supported(int D, int Q, int64_t cls_bstride, int x_bf16, bool bwd) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_POOL_WIDEB"); on = e ? atoi(e) : 1; }      // (EP_POOL_WIDEB=0: the vector-ALU kernels of ep_pool_wide.hip)
  (void)bwd;
  return on && x_bf16 == 1 && D == 32 * WB_KS * WB_NW && Q >= 1 && Q <= 8 && cls_bstride == 0;
}
int wideb_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  const void* fn = bwd ? (const void*)ep_pool_wideb_bwd_kernel : (const void*)ep_pool_wideb_fwd_kernel;
  const size_t lds = bwd ? WB_LDS_BWD : WB_LDS;
  static bool attr_set[2] = {false, false};          // (once per kernel: the call is host-side state, not a stream operation)
  if (!attr_set[bwd ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
    attr_set[bwd ? 1 : 0] = true;
  }
  if (bwd) hipLaunchKernelGGL(ep_pool_wideb_bwd_kernel, dim3(grid), dim3(WB_NW * 64), lds, st, p);
  else hipLaunchKernelGGL(ep_pool_wideb_fwd_kernel, dim3(grid), dim3(WB_NW * 64), lds, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_wideb_bwd_kernel" : "ep_pool_wideb_fwd_kernel");
  return 0;
}

}  // namespace ep
