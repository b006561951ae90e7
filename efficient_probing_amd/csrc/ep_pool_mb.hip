// EP attentive pooling over bf16-STORED tokens on the bf16 matrix cores (gfx950 / CDNA4), fp32 results.
//
// With bf16 token storage a token pass moves half the bytes, and the vector-ALU kernels (widen + 12 k FMA per token)
// become the bound (0.35 of the HBM peak).  Here BOTH contractions of a token tile run on v_mfma_f32_16x16x32_bf16:
//
//   scores   S[t][q]   = sum_d x[t][d] * (cls[q][d]*scale)        A = x tile (bf16, as stored),  B = queries
//   pooling  P^T[d][q] = sum_t x[t][d] * softmax-weight[q][t]     A = x tile^T,                  B = weights
//
// (reference poolings/ep.py:35-44 forward; the backward is the same pair with dP in place of the queries and
// A*(dA-delta) in place of the softmax weights.)  The tokens ARE bf16, so they enter the matrix cores unchanged; the
// fp32 operand of each contraction (queries / dP rows / softmax weights) is split into three bf16 terms
// hi + mid + lo (8 + 8 + 8 mantissa bits, by truncation: the split is exact to 2^-24) and fed as three MFMAs into the
// same fp32 accumulator.  Every bf16 x bf16 product is exact in fp32, so the result is the fp32 contraction of the
// stored values up to summation order -- the contract of the bf16 storage mode (tests/test_gpu_bf16.py) -- at 3/16 of
// the matrix time the fp32 instruction would need.
//
// Two forms.  The default for D in {256, 384, 512, 768, 1024} is the second half of this file (ep_pool_mb2_*: independent
// 4-wave workgroups, two to four per CU, 16-token tiles); the form described here runs D = 1152 and is the fallback
// (EP_POOL_MB2=0) for the others:
// one workgroup of NW = 8 or 12 waves per CU streams whole images through a ring of 32-token tiles filled by LDS-DMA
// (rows XOR-swizzled on the DMA source address).  Wave w owns the D-slice [D/NW*w, D/NW*(w+1)) (32*NK channels) for both
// contractions; D = 32*NK*NW (1152 = 12 waves x 96 channels: three waves per SIMD):
//   1. partial scores of its slice for the two 16-token blocks of the tile (NK k-steps x 3 terms MFMAs each), summed
//      across the waves through LDS; the MFMA D layout of a score block (lane = (query, token group g), 4 tokens in 4
//      registers) is exactly the B-operand slot layout of step 3, so the weights never move between lanes.  With up to
//      8 queries (PK) the two blocks share one 1 KiB record per wave: lanes 8-15 of every row carry block 1 of query
//      lane-8 (one DPP row shift each way), which halves the exchange;
//   2. lazy-max online softmax on 8 values per lane (every wave redundantly: same values, same order);
//   3. pooling of its slice: the A operand (x^T: 8 tokens of one channel per lane) is built from b32 reads of channel
//      PAIRS and two v_perm_b32 per register pair -- the even channels feed one MFMA, the odd ones the next.
// Per 32-token tile and wave: 12 NK MFMAs (36 at D = 768), 2 NK b128 + 8 NK b32 + 16 b128 LDS reads, two barriers.
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"
#include "ep_sidetask.h"
#include "ep_inpass.h"

namespace ep {

typedef __attribute__((address_space(3))) void* mb_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* mb_gptr_t;
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// EP_MB_ABLATE=1 (diagnostic BUILDS only: tools/build_variant.sh -DEP_MB_ABLATE=1) compiles the stage-ablation switches
// of PoolParams.ablate into the forward kernels.  As run-time branches in the shipped kernels they cost 27 - 70 registers
// (D = 768: 256 + 2 spilled against 188; D = 1024: 51 spilled registers, the pass 184 instead of 81 us; D = 384: 142
// against 115, one workgroup per CU fewer) -- measured in round 4, which is why they are compiled out by default.
#ifndef EP_MB_ABLATE
#define EP_MB_ABLATE 0
#endif
constexpr bool MB_ABLATE = EP_MB_ABLATE != 0;
// EP_MB2_PIPE=1 (A/B builds): the software-pipelined tile loop of the two-workgroup form -- the pooling MFMAs of tile it-1
// issued behind the score MFMAs of tile it, between the two barriers of an iteration.  Measured in round 4 (same box,
// alternating runs, 256 x 768 / 197 x 768 / 196 x 384): the step is SLOWER with it, 0.318 - 0.322 against 0.311 - 0.314 ms,
// 0.2985 against 0.289, 0.163 against 0.159 -- the second pass 131 - 135 against 125 - 128 us in the step.  The ring slot of
// tile it-1 is then free only behind the second barrier, so the refill runs one iteration ahead of its use instead of
// almost two, and the two workgroups of a CU already cover each other's chains.  Kept as the measured alternative; off.
#ifndef EP_MB2_PIPE
#define EP_MB2_PIPE 0
#endif

constexpr int MB_TT = 32;             // tokens per tile (two 16-token MFMA blocks)
constexpr float MB_LOG2E = 1.4426950408889634f;
constexpr float MB_LAZY_MAX_THR = 12.0f;

template <int NK, int NW, bool PK>    // D = 32 * NK * NW;  PK: packed score exchange (Q <= 8)
struct MbCfg {
  static constexpr int D = 32 * NK * NW;
  static constexpr int ROWB = 2 * D;                 // bytes per token row
  static constexpr int NCH = D / 8;                  // 16-byte chunks per row (multiple of 16)
  static constexpr int SLOT = MB_TT * ROWB;
  static constexpr int KDMA = SLOT / (NW * 1024);    // 1 KiB DMA pieces per wave per tile
  static constexpr int SPART = (PK ? 1 : 2) * NW * 1024;   // partial score records [block][wave][lane] f4
  static constexpr int LDS_TOTAL = 160 * 1024;
  static constexpr int NSLOT = ((LDS_TOTAL - SPART) / SLOT) > 3 ? 3 : ((LDS_TOTAL - SPART) / SLOT);
  static constexpr bool VALID = NSLOT >= 2 && NCH % 16 == 0 && SLOT % (NW * 1024) == 0;
};

__device__ __forceinline__ void mb_wait_vmcnt(int n) {
#define EP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    EP_W(0) EP_W(1) EP_W(2) EP_W(3) EP_W(4) EP_W(5) EP_W(6) EP_W(7) EP_W(8) EP_W(9)
    EP_W(10) EP_W(11) EP_W(12) EP_W(13) EP_W(14) EP_W(15) EP_W(16)
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
#undef EP_W
}
__device__ __forceinline__ void mb_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// combine a per-lane value over the 4 lanes that share a query (lane, lane^16, lane^32, lane^48)
__device__ __forceinline__ float mb_q4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float mb_q4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// fp32 -> three bf16 terms by truncation (hi + mid + lo == v to 2^-24 |v|); the terms are returned as fp32 bit
// patterns whose low 16 bits are zero
__device__ __forceinline__ void mb_split3(float v, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(v) & 0xffff0000u;
  const float r1 = v - __uint_as_float(h);                    // exact
  m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);                   // exact
  l = __float_as_uint(r2) & 0xffff0000u;
}
// two truncated terms -> one register of two bf16 (element 0 = a in the low half)
__device__ __forceinline__ unsigned mb_pack_hi(unsigned a, unsigned b) {
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
// eight fp32 values -> the three bf16x8 operands
__device__ __forceinline__ void mb_split8(const float (&v)[8], u4 (&t)[3]) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) mb_split3(v[e], h[e], m[e], l[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    t[0][e] = mb_pack_hi(h[2 * e], h[2 * e + 1]);
    t[1][e] = mb_pack_hi(m[2 * e], m[2 * e + 1]);
    t[2][e] = mb_pack_hi(l[2 * e], l[2 * e + 1]);
  }
}
__device__ __forceinline__ f4 mb_mfma(u4 a, u4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// LDS position p = (t, c') of a tile holds source chunk c' ^ (t & 15) of row t
template <int NCH, int NW, int KDMA>
__device__ __forceinline__ void mb_source_offsets(int w, int lane, unsigned (&soff)[KDMA]) {
#pragma unroll
  for (int jj = 0; jj < KDMA; ++jj) {
    const int pos = (w + NW * jj) * 64 + lane;
    const int t = pos / NCH, c = pos - t * NCH;
    soff[jj] = (unsigned)(t * (16 * NCH) + ((c ^ (t & 15)) << 4));
  }
}
template <int NW, int KDMA>
__device__ __forceinline__ void mb_dma_tile(const char* src, unsigned limit, char* slot, int w,
                                            const unsigned (&soff)[KDMA]) {
#pragma unroll
  for (int jj = 0; jj < KDMA; ++jj) {
    const unsigned off = soff[jj] < limit ? soff[jj] : limit;
    __builtin_amdgcn_global_load_lds((mb_gptr_t)(src + off), (mb_lds_ptr_t)(slot + (w + NW * jj) * 1024), 16, 0,
                                     EP_DMA_AUX);
  }
}

// step 1: (2 x 16 tokens) x 16 queries over this wave's D-slice -> LDS scratch; `mid` (the ring refill) is issued in the
// shadow of the first MFMAs.  A operand: lane (i = token of the block, g): chunk 4*NK*w + 4*ks + g of row i.
template <int NK, int NW, bool PK, typename F>
__device__ __forceinline__ void mb_scores(const char* tile, const int (&aoff)[NK], const u4 (&bq)[NK][3], char* spart,
                                          int w, int lane, F&& mid) {
  constexpr int ROWB = 64 * NK * NW;
  u4 xa[2][NK];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) xa[blk][ks] = *reinterpret_cast<const u4*>(tile + blk * 16 * ROWB + aoff[ks]);
  f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
    for (int term = 2; term >= 0; --term) {                    // small terms first
      acc[0] = mb_mfma(xa[0][ks], bq[ks][term], acc[0]);
      acc[1] = mb_mfma(xa[1][ks], bq[ks][term], acc[1]);
    }
    if (ks == 0) mid();
  }
  if (PK) {                                                    // lanes 8-15 of a row: block 1 of query lane - 8
    const bool up = (lane & 8) != 0;
    const float s0 = dpp_f<0x118>(acc[1].x), s1 = dpp_f<0x118>(acc[1].y);   // row_shr:8 (all lanes active)
    const float s2 = dpp_f<0x118>(acc[1].z), s3 = dpp_f<0x118>(acc[1].w);
    f4 v;
    v.x = up ? s0 : acc[0].x; v.y = up ? s1 : acc[0].y; v.z = up ? s2 : acc[0].z; v.w = up ? s3 : acc[0].w;
    *reinterpret_cast<f4*>(spart + (w * 64 + lane) * 16) = v;
  } else {
    *reinterpret_cast<f4*>(spart + ((0 * NW + w) * 64 + lane) * 16) = acc[0];
    *reinterpret_cast<f4*>(spart + ((1 * NW + w) * 64 + lane) * 16) = acc[1];
  }
}
// full scores of (query j = lane & 15, tokens 16*blk + 4*g + r): the same lane slot of every wave's record, fixed order
template <int NW, bool PK>
__device__ __forceinline__ void mb_gather(const char* spart, int lane, float (&s)[8]) {
  if (PK) {
    f4 v = *reinterpret_cast<const f4*>(spart + lane * 16);
#pragma unroll
    for (int ws = 1; ws < NW; ++ws) v += *reinterpret_cast<const f4*>(spart + (ws * 64 + lane) * 16);
    s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;            // lanes 0-7 of a row: block 0 of query j
    s[4] = dpp_f<0x108>(v.x); s[5] = dpp_f<0x108>(v.y);        // row_shl:8: block 1 from lane + 8
    s[6] = dpp_f<0x108>(v.z); s[7] = dpp_f<0x108>(v.w);
  } else {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f4 v = *reinterpret_cast<const f4*>(spart + ((blk * NW) * 64 + lane) * 16);
#pragma unroll
      for (int ws = 1; ws < NW; ++ws) v += *reinterpret_cast<const f4*>(spart + ((blk * NW + ws) * 64 + lane) * 16);
      s[4 * blk + 0] = v.x; s[4 * blk + 1] = v.y; s[4 * blk + 2] = v.z; s[4 * blk + 3] = v.w;
    }
  }
}
// step 3: for each group of 32 channels dg of the slice: accE[dg] (rows = even channels) and accO[dg] (odd channels),
// 16 channels x 16 queries each, += x_tile^T * wgt.  Operand slot (g, e) <-> token 16*(e>>2) + 4*g + (e&3) on both sides.
// Lane (i, g) reads the channel pair (2i, 2i+1) of the group: bytes 4*i of the 64-byte segment at chunk
// c0 = 4*NK*w + 4*dg, i.e. chunk c0 + (i>>2), stored at chunk position (c0 + (i>>2)) ^ (4g + r) of row t
// (conflict-free: the four g land in four different 64-byte segments of a 256-byte window).
template <int NK, int NW>
__device__ __forceinline__ void mb_pool(const char* tile, const int (&poff)[4], const int (&pseg)[NK],
                                        const float (&wgt)[8], f4 (&accE)[NK], f4 (&accO)[NK]) {
  constexpr int ROWB = 64 * NK * NW;
  u4 bw[3];
  mb_split8(wgt, bw);
  unsigned xr[NK][8];
#pragma unroll
  for (int dg = 0; dg < NK; ++dg)
#pragma unroll
    for (int e = 0; e < 8; ++e)
      xr[dg][e] = *reinterpret_cast<const unsigned*>(tile + (e >> 2) * 16 * ROWB + poff[e & 3] + pseg[dg]);
#pragma unroll
  for (int dg = 0; dg < NK; ++dg) {
    u4 ae, ao;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ae[e] = __builtin_amdgcn_perm(xr[dg][2 * e + 1], xr[dg][2 * e], 0x05040100u);   // (lo of 2e, lo of 2e+1)
      ao[e] = __builtin_amdgcn_perm(xr[dg][2 * e + 1], xr[dg][2 * e], 0x07060302u);   // (hi of 2e, hi of 2e+1)
    }
#pragma unroll
    for (int term = 2; term >= 0; --term) {
      accE[dg] = mb_mfma(ae, bw[term], accE[dg]);
      accO[dg] = mb_mfma(ao, bw[term], accO[dg]);
    }
  }
}
// per-lane address parts of the pooling A operand: row 4g + r, chunk-in-segment (i>>2) ^ r, channel pair i & 3;
// 64-byte segment (NK*w + dg) ^ g of the row
template <int NK, int NW>
__device__ __forceinline__ void mb_pool_offsets(int w, int i, int g, int (&poff)[4], int (&pseg)[NK]) {
  constexpr int ROWB = 64 * NK * NW;
#pragma unroll
  for (int r = 0; r < 4; ++r) poff[r] = (4 * g + r) * ROWB + (((i >> 2) ^ r) << 4) + 4 * (i & 3);
#pragma unroll
  for (int dg = 0; dg < NK; ++dg) pseg[dg] = ((NK * w + dg) ^ g) << 6;
}
// accumulators of one 32-channel group -> 8 consecutive channels per lane: d = 32*dg + 8*g + {0..7}
__device__ __forceinline__ void mb_store8(float* dst, f4 e, f4 o, float f) {
  *reinterpret_cast<f4*>(dst) = f4{e.x * f, o.x * f, e.y * f, o.y * f};
  *reinterpret_cast<f4*>(dst + 4) = f4{e.z * f, o.z * f, e.w * f, o.w * f};
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int NK, int NW, bool PK>
__global__ __launch_bounds__(NW * 64) void ep_pool_mb_fwd_kernel(PoolParams p) {
  using C = MbCfg<NK, NW, PK>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT, KDMA = C::KDMA, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + MB_TT - 1) / MB_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  const int j = lane & 15, g = lane >> 4;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x);

  // B operand of the score MFMAs: queries pre-scaled like the reference (ep.py:39), split into three bf16 terms
  u4 bq[NK][3];
  int aoff[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < Q) {
      const float* src = p.cls + (int64_t)j * D + 32 * NK * w + 32 * ks + 8 * g;
      const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
      v[0] = a.x * p.scale; v[1] = a.y * p.scale; v[2] = a.z * p.scale; v[3] = a.w * p.scale;
      v[4] = b.x * p.scale; v[5] = b.y * p.scale; v[6] = b.z * p.scale; v[7] = b.w * p.scale;
    }
    mb_split8(v, bq[ks]);
    aoff[ks] = j * ROWB + (((4 * NK * w + 4 * ks + g) ^ j) << 4);
  }
  unsigned soff[KDMA];
  mb_source_offsets<NCH, NW, KDMA>(w, lane, soff);
  int poff[4], pseg[NK];
  mb_pool_offsets<NK, NW>(w, j, g, poff, pseg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, wg));
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * MB_TT;
      const unsigned limit = (unsigned)((left < MB_TT ? left : MB_TT) * ROWB - 16);
      mb_dma_tile<NW, KDMA>(psrc, limit, ring + pslot * SLOT, w, soff);
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
      } else {
        psrc += SLOT;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 accE[NK], accO[NK];
  float m_j = -INFINITY, mL_j = -INFINITY, lsum = 0.f;     // per lane: running max / partial sum of query j
  int cimg = 0, ctile = 0, cslot = 0;
  for (int it = 0; it < n_items; ++it) {
    mb_wait_vmcnt((pi - 1 - it) * KDMA);
    mb_barrier();                                   // tile `it` landed everywhere; the slot of tile it-1 is free
    const int b = wg + cimg * G;
    const int n0 = ctile * MB_TT;
    const int nvalid = (N - n0) < MB_TT ? (N - n0) : MB_TT;
    const char* tile = ring + cslot * SLOT;
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    if (ctile == 0) {
      m_j = -INFINITY; mL_j = -INFINITY; lsum = 0.f;
#pragma unroll
      for (int dg = 0; dg < NK; ++dg) { accE[dg] = f4{0.f, 0.f, 0.f, 0.f}; accO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }
    }
    if (MB_ABLATE && p.ablate == 1) { produce(); continue; }     // diagnostic: ring only
    mb_scores<NK, NW, PK>(tile, aoff, bq, spart, w, lane, produce);
    float sc[8], ue[8];
    if (MB_ABLATE && p.ablate == 2) {               // diagnostic: no exchange
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] = 0.f;
    } else {
      mb_barrier();                                 // all partial score blocks are in the scratch
      mb_gather<NW, PK>(spart, lane, sc);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ue[e] = (16 * (e >> 2) + 4 * g + (e & 3)) < nvalid ? sc[e] : -INFINITY;
      mx = fmaxf(mx, ue[e]);
    }
    if (__builtin_amdgcn_ballot_w64(mx > m_j + MB_LAZY_MAX_THR) != 0ull) {    // rare
      const float mn = fmaxf(m_j, mb_q4_max(mx));
      const float f = __builtin_amdgcn_exp2f((m_j - mn) * MB_LOG2E);           // m = -inf -> 0
      m_j = mn; mL_j = mn * MB_LOG2E;
      lsum *= f;
#pragma unroll
      for (int dg = 0; dg < NK; ++dg) { accE[dg] *= f; accO[dg] *= f; }       // my column is query j
    }
    float wgt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wgt[e] = __builtin_amdgcn_exp2f(fmaf(ue[e], MB_LOG2E, -mL_j));          // invalid tokens: 0
      lsum += wgt[e];
    }
    // every wave holds the same scores: wave w writes token group (w >> 1) & 3 of block w & 1 (waves 8-11: nothing)
    if (w < 8 && g == (w >> 1) && j < Q) {
      const int blk = w & 1;
      const int t0 = 16 * blk + 4 * g;
      float* Srow = p.S + ((int64_t)b * QS + j) * N + n0 + t0;
      const float s0 = blk ? sc[4] : sc[0], s1 = blk ? sc[5] : sc[1], s2 = blk ? sc[6] : sc[2], s3 = blk ? sc[7] : sc[3];
      if (n4) {
        if (t0 < nvalid) *reinterpret_cast<f4*>(Srow) = f4{s0, s1, s2, s3};
      } else {
        if (t0 + 0 < nvalid) Srow[0] = s0;
        if (t0 + 1 < nvalid) Srow[1] = s1;
        if (t0 + 2 < nvalid) Srow[2] = s2;
        if (t0 + 3 < nvalid) Srow[3] = s3;
      }
    }
    if (!MB_ABLATE || p.ablate != 3) mb_pool<NK, NW>(tile, poff, pseg, wgt, accE, accO);
    if (ctile == tiles_per_img - 1) {
      const float l = mb_q4_sum(lsum);
      const float inv = 1.0f / l;
      if (j < Q) {
        float* Pq = p.P + ((int64_t)b * QS + j) * D + 32 * NK * w + 8 * g;
#pragma unroll
        for (int dg = 0; dg < NK; ++dg) mb_store8(Pq + 32 * dg, accE[dg], accO[dg], inv);
        if (w == 0 && g == 0) {
          const f4 rec = {m_j, l, 0.f, 0.f};
          *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + j) * 4) = rec;
        }
      }
      ctile = 0; ++cimg;
    } else {
      ++ctile;
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward.  Per image: the Q rows of dP[b] (this wave's slice, split into bf16 terms) and the ML row of query j are
// fetched into registers one image ahead; the saved scores of a tile (8 per lane, the lane's operand slots) one tile
// ahead.  These plain loads are issued BEFORE the ring refill of their iteration, so the counted wait at the top of
// the next iteration (which leaves only the newest tile's DMA outstanding) covers them.
// ---------------------------------------------------------------------------------------
template <int NK, int NW, bool PK>
__global__ __launch_bounds__(NW * 64) void ep_pool_mb_bwd_kernel(PoolParams p) {
  using C = MbCfg<NK, NW, PK>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT, KDMA = C::KDMA, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + MB_TT - 1) / MB_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  const int j = lane & 15, g = lane >> 4;
  const int jq = j < Q ? j : Q - 1;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x);

  f4 gE[NK], gO[NK];
#pragma unroll
  for (int dg = 0; dg < NK; ++dg) { gE[dg] = f4{0.f, 0.f, 0.f, 0.f}; gO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }

  if (n_items > 0) {
    int aoff[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) aoff[ks] = j * ROWB + (((4 * NK * w + 4 * ks + g) ^ j) << 4);
    unsigned soff[KDMA];
    mb_source_offsets<NCH, NW, KDMA>(w, lane, soff);
    int poff[4], pseg[NK];
    mb_pool_offsets<NK, NW>(w, j, g, poff, pseg);

    // header of image `img` (index into this workgroup's images) -> registers
    f4 hq[NK][2], hml;
    auto load_header = [&](int img) {
      const int b = wg + img * G;
      const float* src = p.dP + ((int64_t)b * QS + jq) * D + 32 * NK * w + 8 * g;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        hq[ks][0] = *reinterpret_cast<const f4*>(src + 32 * ks);
        hq[ks][1] = *reinterpret_cast<const f4*>(src + 32 * ks + 4);
      }
      hml = *reinterpret_cast<const f4*>(p.ML + ((int64_t)b * QS + jq) * 4);
    };
    // saved scores of tile (img, tile): S[b, j, n0 + 16 blk + 4 g + r]
    float sv[8];
    auto load_scores = [&](int img, int tl) {
      const int b = wg + img * G;
      const float* Srow = p.S + ((int64_t)b * QS + jq) * N;
      const int n0 = tl * MB_TT;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        int t0 = n0 + 16 * blk + 4 * g;
        if (n4) {
          t0 = t0 < N ? t0 : N - 4;
          const f4 v = *reinterpret_cast<const f4*>(Srow + t0);
          sv[4 * blk] = v.x; sv[4 * blk + 1] = v.y; sv[4 * blk + 2] = v.z; sv[4 * blk + 3] = v.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[4 * blk + r] = Srow[(t0 + r) < N ? (t0 + r) : N - 1];
        }
      }
    };

    load_header(0);
    load_scores(0, 0);
    int pi = 0, pimg = 0, ptile = 0, pslot = 0;
    const char* psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, wg));
    auto produce = [&]() {
      if (pi < n_items) {
        const int left = N - ptile * MB_TT;
        const unsigned limit = (unsigned)((left < MB_TT ? left : MB_TT) * ROWB - 16);
        mb_dma_tile<NW, KDMA>(psrc, limit, ring + pslot * SLOT, w, soff);
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++ptile == tiles_per_img) {
          ptile = 0; ++pimg;
          psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
        } else {
          psrc += SLOT;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    u4 bq[NK][3];
    float mL_j = 0.f, il_j = 0.f, dl_j = 0.f;
    int cimg = 0, ctile = 0, cslot = 0;
    for (int it = 0; it < n_items; ++it) {
      mb_wait_vmcnt((pi - 1 - it) * KDMA);
      mb_barrier();
      const int n0 = ctile * MB_TT;
      const int nvalid = (N - n0) < MB_TT ? (N - n0) : MB_TT;
      const char* tile = ring + cslot * SLOT;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if (ctile == 0) {                              // new image: its header is in hq / hml
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          float v[8] = {hq[ks][0].x, hq[ks][0].y, hq[ks][0].z, hq[ks][0].w, hq[ks][1].x, hq[ks][1].y, hq[ks][1].z, hq[ks][1].w};
          if (j >= Q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
          }
          mb_split8(v, bq[ks]);
        }
        mL_j = hml.x * MB_LOG2E; il_j = 1.0f / hml.y; dl_j = hml.z;
        if (cimg + 1 < n_img) load_header(cimg + 1);
      }
      float cur[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cur[e] = sv[e];
      {                                              // scores of the next tile (before the refill: see above)
        int nimg = cimg, ntile = ctile + 1;
        if (ntile == tiles_per_img) { ntile = 0; ++nimg; }
        if (nimg < n_img) load_scores(nimg, ntile);
      }
      mb_scores<NK, NW, PK>(tile, aoff, bq, spart, w, lane, produce);      // dA partial blocks
      mb_barrier();
      float u[8], wgt[8];
      mb_gather<NW, PK>(spart, lane, u);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = __builtin_amdgcn_exp2f(fmaf(cur[e], MB_LOG2E, -mL_j)) * il_j;
        wgt[e] = ((16 * (e >> 2) + 4 * g + (e & 3)) < nvalid && j < Q) ? a * (u[e] - dl_j) : 0.f;
      }
      mb_pool<NK, NW>(tile, poff, pseg, wgt, gE, gO);
      if (++ctile == tiles_per_img) { ctile = 0; ++cimg; }
    }
  }
  if (j < Q) {
    float* Gq = p.Gpart + ((int64_t)wg * Q + j) * D + 32 * NK * w + 8 * g;
#pragma unroll
    for (int dg = 0; dg < NK; ++dg) mb_store8(Gq + 32 * dg, gE[dg], gO[dg], 1.0f);
  }
}

// =======================================================================================
// Two-workgroup form (D = 128*NK <= 768): 4-wave workgroups, TWO per CU, 16-token tiles.  Each workgroup has its own
// ring and its own barriers, so one workgroup's rendezvous and LDS latencies are covered by the other's arithmetic; a
// wave owns 32*NK channels (192 at D = 768), which halves the redundant softmax / operand-split work per token.  The
// pooling contraction runs on v_mfma_f32_16x16x16_bf16 (K = the 16 tokens of the tile): operand slot (g, e) <-> token
// 4g + e, which is the D layout of the score block.
// =======================================================================================
typedef short s4v __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr int MB2_TT = 16, MB2_NW = 4;

template <int NK, int NS = 3>
struct Mb2Cfg {
  static constexpr int D = 128 * NK;
  static constexpr int ROWB = 2 * D;
  static constexpr int NCH = D / 8;
  static constexpr int SLOT = MB2_TT * ROWB;
  static constexpr int KDMA = SLOT / (MB2_NW * 1024);   // = NK
  static constexpr int SPART = MB2_NW * 1024;
  static constexpr int NSLOT = NS;                      // ring slots (2 where three do not leave room for two workgroups)
  static constexpr int LDS = NSLOT * SLOT + SPART;      // 77824 at D = 768: two workgroups per CU
};

__device__ __forceinline__ f4 mb2_mfma16(u2 a, u2 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, a), __builtin_bit_cast(s4v, b), c, 0, 0, 0);
}
// NT (mb2_scores / mb2_pool / the mb2 kernels): terms of the fp32 operand that are multiplied -- 3: fp32 results; 1: the AMP-bf16
// arithmetic mode (PoolParams.nterms: bf16(query) x token and bf16(weight) x token as ONE product with fp32 accumulation, what
// the reference's q @ k^T and attn @ v do under --amp bfloat16, poolings/ep.py:41-44 inside engine_finetune.py:52-55's autocast)
template <int NT = 3>
__device__ __forceinline__ void mb2_split4(const float (&v)[4], u2 (&t)[3]) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) mb_split3(v[e], h[e], m[e], l[e]);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    t[0][e] = mb_pack_hi(h[2 * e], h[2 * e + 1]);
    if constexpr (NT == 3) {
      t[1][e] = mb_pack_hi(m[2 * e], m[2 * e + 1]);
      t[2][e] = mb_pack_hi(l[2 * e], l[2 * e + 1]);
    }
  }
}
// partial scores of the 16-token tile over this wave's slice -> its record in the scratch
template <int NK, int NT = 3, typename F>
__device__ __forceinline__ void mb2_scores(const char* tile, const int (&aoff)[NK], const u4 (&bq)[NK][3], char* spart,
                                           int w, int lane, F&& mid) {
  u4 xa[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) xa[ks] = *reinterpret_cast<const u4*>(tile + aoff[ks]);
  f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
    for (int term = NT - 1; term >= 0; --term) acc[ks & 1] = mb_mfma(xa[ks], bq[ks][term], acc[ks & 1]);
    if (ks == 0) mid();
  }
  *reinterpret_cast<f4*>(spart + (w * 64 + lane) * 16) = acc[0] + acc[1];
}
__device__ __forceinline__ void mb2_gather(const char* spart, int lane, float (&s)[4]) {
  f4 v = *reinterpret_cast<const f4*>(spart + lane * 16);
#pragma unroll
  for (int ws = 1; ws < MB2_NW; ++ws) v += *reinterpret_cast<const f4*>(spart + (ws * 64 + lane) * 16);
  s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
}
template <int NK, int NT = 3>
__device__ __forceinline__ void mb2_pool(const char* tile, const int (&poff)[4], const int (&pseg)[NK],
                                         const float (&wgt)[4], f4 (&accE)[NK], f4 (&accO)[NK]) {
  u2 bw[3];
  mb2_split4<NT>(wgt, bw);
  unsigned xr[NK][4];
#pragma unroll
  for (int dg = 0; dg < NK; ++dg)
#pragma unroll
    for (int e = 0; e < 4; ++e) xr[dg][e] = *reinterpret_cast<const unsigned*>(tile + poff[e] + pseg[dg]);
#pragma unroll
  for (int dg = 0; dg < NK; ++dg) {
    u2 ae, ao;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ae[e] = __builtin_amdgcn_perm(xr[dg][2 * e + 1], xr[dg][2 * e], 0x05040100u);
      ao[e] = __builtin_amdgcn_perm(xr[dg][2 * e + 1], xr[dg][2 * e], 0x07060302u);
    }
#pragma unroll
    for (int term = NT - 1; term >= 0; --term) {
      accE[dg] = mb2_mfma16(ae, bw[term], accE[dg]);
      accO[dg] = mb2_mfma16(ao, bw[term], accO[dg]);
    }
  }
}

template <int NK, int NS, int NT = 3>
__global__ __launch_bounds__(MB2_NW * 64, 2) void ep_pool_mb2_fwd_kernel(PoolParams p) {
  using C = Mb2Cfg<NK, NS>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT, KDMA = C::KDMA, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + MB2_TT - 1) / MB2_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (wg == 0 && p.ip_zero)                         // the second pass's in-pass counters (ep_inpass.h): zero before it starts
    for (int t = threadIdx.x; t < p.ip_nzero; t += MB2_NW * 64) p.ip_zero[t] = 0;
  if (n_items <= 0) return;
  const int j = lane & 15, g = lane >> 4;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x);

  u4 bq[NK][3];
  int aoff[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < Q) {
      const float* src = p.cls + (int64_t)j * D + 32 * NK * w + 32 * ks + 8 * g;
      const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
      v[0] = a.x * p.scale; v[1] = a.y * p.scale; v[2] = a.z * p.scale; v[3] = a.w * p.scale;
      v[4] = b.x * p.scale; v[5] = b.y * p.scale; v[6] = b.z * p.scale; v[7] = b.w * p.scale;
    }
    mb_split8(v, bq[ks]);
    aoff[ks] = j * ROWB + (((4 * NK * w + 4 * ks + g) ^ j) << 4);
  }
  unsigned soff[KDMA];
  mb_source_offsets<NCH, MB2_NW, KDMA>(w, lane, soff);
  int poff[4], pseg[NK];
  mb_pool_offsets<NK, MB2_NW>(w, j, g, poff, pseg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, wg));
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * MB2_TT;
      const unsigned limit = (unsigned)((left < MB2_TT ? left : MB2_TT) * ROWB - 16);
      mb_dma_tile<MB2_NW, KDMA>(psrc, limit, ring + pslot * SLOT, w, soff);
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
      } else {
        psrc += SLOT;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 accE[NK], accO[NK];
  float m_j = -INFINITY, mL_j = -INFINITY, lsum = 0.f;
  int cimg = 0, ctile = 0, cslot = 0;
  if constexpr (NS >= 3 && EP_MB2_PIPE) {
    // SOFTWARE-PIPELINED tile loop (round 4): the pooling MFMAs of tile it-1 are issued between the two barriers of
    // iteration `it`, right behind the score MFMAs of tile it -- two independent chains, so one's LDS-read and
    // matrix-pipe latencies are covered by the other instead of adding up (no unit of the CU is more than ~40 % busy
    // in this pass: it is bound by the serial chain of a tile).  The slot of tile it-1 is therefore free only after the
    // SECOND barrier of iteration it, where the ring is refilled (tile it+2: three slots hold it-1 | it | it+1).
    // The raw scores of tile it-1 are stored in iteration `it` as well, IN FRONT of the ring refill: vmcnt retires in
    // order, so a store issued behind the newest tile's copies would make the counted wait at the top of the next
    // iteration wait for the first of those copies too.
    float pwgt[4] = {0.f, 0.f, 0.f, 0.f}, psc[4] = {0.f, 0.f, 0.f, 0.f};
    const char* ptile_ = ring;
    bool plast = false; int pb = 0, pn0 = 0, pnvalid = 0;
    for (int it = 0; it <= n_items; ++it) {
      const bool live = it < n_items;
      if (live) mb_wait_vmcnt((pi - 1 - it) * KDMA);
      mb_barrier();                                 // tile `it` landed everywhere; everyone is past gather(it-1)
      const int b = wg + cimg * G;
      const int n0 = ctile * MB2_TT;
      const int nvalid = (N - n0) < MB2_TT ? (N - n0) : MB2_TT;
      const char* tile = ring + cslot * SLOT;
      if (live) mb2_scores<NK, NT>(tile, aoff, bq, spart, w, lane, [] {});
      if (it > 0) {
        mb2_pool<NK, NT>(ptile_, poff, pseg, pwgt, accE, accO);
        if (w == ((it - 1) & 3) && j < Q) {         // every wave holds the same scores: one writes the tile
          float* Srow = p.S + ((int64_t)pb * QS + j) * N + pn0 + 4 * g;
          if (n4) {
            if (4 * g < pnvalid) *reinterpret_cast<f4*>(Srow) = f4{psc[0], psc[1], psc[2], psc[3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (4 * g + r < pnvalid) Srow[r] = psc[r];
          }
        }
        if (plast) {                                // tile it-1 closed its image: normalise and store
          const float l = mb_q4_sum(lsum);
          const float inv = 1.0f / l;
          if (j < Q) {
            float* Pq = p.P + ((int64_t)pb * QS + j) * D + 32 * NK * w + 8 * g;
#pragma unroll
            for (int dg = 0; dg < NK; ++dg) mb_store8(Pq + 32 * dg, accE[dg], accO[dg], inv);
            if (w == 0 && g == 0) {
              const f4 rec = {m_j, l, 0.f, 0.f};
              *reinterpret_cast<f4*>(p.ML + ((int64_t)pb * QS + j) * 4) = rec;
            }
          }
        }
      }
      if (!live) break;
      mb_barrier();                                 // all partial score blocks are in the scratch; slot of tile it-1 is free
      produce();
      if (ctile == 0) {
        m_j = -INFINITY; mL_j = -INFINITY; lsum = 0.f;
#pragma unroll
        for (int dg = 0; dg < NK; ++dg) { accE[dg] = f4{0.f, 0.f, 0.f, 0.f}; accO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }
      }
      float sc[4], ue[4];
      mb2_gather(spart, lane, sc);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ue[e] = (4 * g + e) < nvalid ? sc[e] : -INFINITY;
        mx = fmaxf(mx, ue[e]);
      }
      if (__builtin_amdgcn_ballot_w64(mx > m_j + MB_LAZY_MAX_THR) != 0ull) {
        const float mn = fmaxf(m_j, mb_q4_max(mx));
        const float f = __builtin_amdgcn_exp2f((m_j - mn) * MB_LOG2E);
        m_j = mn; mL_j = mn * MB_LOG2E;
        lsum *= f;
#pragma unroll
        for (int dg = 0; dg < NK; ++dg) { accE[dg] *= f; accO[dg] *= f; }      // holds the tiles up to it-1 of this image
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pwgt[e] = __builtin_amdgcn_exp2f(fmaf(ue[e], MB_LOG2E, -mL_j));
        lsum += pwgt[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) psc[e] = sc[e];
      ptile_ = tile; pb = b; pn0 = n0; pnvalid = nvalid;
      plast = ctile == tiles_per_img - 1;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if (plast) { ctile = 0; ++cimg; } else { ++ctile; }
    }
    return;
  }
  for (int it = 0; it < n_items; ++it) {
    mb_wait_vmcnt((pi - 1 - it) * KDMA);
    mb_barrier();
    const int b = wg + cimg * G;
    const int n0 = ctile * MB2_TT;
    const int nvalid = (N - n0) < MB2_TT ? (N - n0) : MB2_TT;
    const char* tile = ring + cslot * SLOT;
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    if (ctile == 0) {
      m_j = -INFINITY; mL_j = -INFINITY; lsum = 0.f;
#pragma unroll
      for (int dg = 0; dg < NK; ++dg) { accE[dg] = f4{0.f, 0.f, 0.f, 0.f}; accO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }
    }
    if (MB_ABLATE && p.ablate == 1) { produce(); continue; }     // diagnostic: ring only
    mb2_scores<NK, NT>(tile, aoff, bq, spart, w, lane, produce);
    float sc[4], ue[4];
    if (MB_ABLATE && p.ablate == 2) {               // diagnostic: no exchange
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = 0.f;
    } else {
      mb_barrier();
      mb2_gather(spart, lane, sc);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ue[e] = (4 * g + e) < nvalid ? sc[e] : -INFINITY;
      mx = fmaxf(mx, ue[e]);
    }
    if (__builtin_amdgcn_ballot_w64(mx > m_j + MB_LAZY_MAX_THR) != 0ull) {
      const float mn = fmaxf(m_j, mb_q4_max(mx));
      const float f = __builtin_amdgcn_exp2f((m_j - mn) * MB_LOG2E);
      m_j = mn; mL_j = mn * MB_LOG2E;
      lsum *= f;
#pragma unroll
      for (int dg = 0; dg < NK; ++dg) { accE[dg] *= f; accO[dg] *= f; }
    }
    float wgt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wgt[e] = __builtin_amdgcn_exp2f(fmaf(ue[e], MB_LOG2E, -mL_j));
      lsum += wgt[e];
    }
    if (w == (it & 3) && j < Q && (!MB_ABLATE || p.ablate != 4)) {  // every wave holds the same scores: one writes the tile
      float* Srow = p.S + ((int64_t)b * QS + j) * N + n0 + 4 * g;
      if (n4) {
        if (4 * g < nvalid) *reinterpret_cast<f4*>(Srow) = f4{sc[0], sc[1], sc[2], sc[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < nvalid) Srow[r] = sc[r];
      }
    }
    if (!MB_ABLATE || p.ablate != 3) mb2_pool<NK, NT>(tile, poff, pseg, wgt, accE, accO);
    if (ctile == tiles_per_img - 1) {
      const float l = mb_q4_sum(lsum);
      const float inv = 1.0f / l;
      if (j < Q) {
        float* Pq = p.P + ((int64_t)b * QS + j) * D + 32 * NK * w + 8 * g;
#pragma unroll
        for (int dg = 0; dg < NK; ++dg) mb_store8(Pq + 32 * dg, accE[dg], accO[dg], inv);
        if (w == 0 && g == 0) {
          const f4 rec = {m_j, l, 0.f, 0.f};
          *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + j) * 4) = rec;
        }
      }
      ctile = 0; ++cimg;
    } else {
      ++ctile;
    }
  }
}

// SIDE: the launch carries the step's weight-gradient contractions, the bias column sum and the statistics fold as extra
// workgroups behind the pooling grid (ep_sidetask.h: the dispatcher places them as pooling workgroups retire, i.e. into
// the tail of the pass -- no second stream, no cross-queue events).  A template flag because the contraction tile's
// registers must not push the D <= 384 pooling kernels below four workgroups per CU when nothing rides along.
template <int NK, int NS, bool SIDE, int NT = 3>
__global__ __launch_bounds__(MB2_NW * 64, 2) void ep_pool_mb2_bwd_kernel(PoolParams p, SideTasks side) {
  using C = Mb2Cfg<NK, NS>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT, KDMA = C::KDMA, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int G = SIDE ? (int)gridDim.x - side.total : (int)gridDim.x;      // pooling workgroups
  const int wg = blockIdx.x;
  if constexpr (SIDE) {
    if (wg >= G) { run_side_task(side, wg - G, lds); return; }
  }
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + MB2_TT - 1) / MB2_TT;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  const int j = lane & 15, g = lane >> 4;
  const int jq = j < Q ? j : Q - 1;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x);
  // ---- in-pass dP (ep_inpass.h; Q = 8, D = 256 KT, whole row blocks of 32 images, the whole pooling grid resident --
  // host-checked, ep_pool.hip: pool_inpass_mask): this workgroup's tasks first, then the wait for the row blocks of its own
  // images, then the stream.  The dP rows are read with sc1 loads below (written by other workgroups of this launch).
  constexpr bool IPOK = SIDE && (NK == 2 || NK == 4 || NK == 6);
  if constexpr (IPOK) {
    if (wg == 0 && p.ip_zero)
      for (int t = threadIdx.x; t < p.ip_nzero; t += MB2_NW * 64) p.ip_zero[t] = 0;
    if (p.ip_dy) {
      const int R = (p.B + G - 1) / G;
      const int nfull = p.B - (R - 1) * G;                   // workgroups with R images
      const int nh = G - nfull;                              // helpers
      auto run = [&](int b) {
        // (a CALL: inlined, the task's ~90 staging registers pushed this kernel's token loop into scratch)
        ip_dp_task_call<NK / 2>(p.ip_dy, p.ip_Wv, const_cast<float*>(p.dP), p.B, b, lds);
        ip_arrive(p.ip_dcnt + (b >> 5) * IP_CNT_STRIDE);
      };
      run(wg);
      if (nh == 0) { for (int b = wg + G; b < p.B; b += G) run(b); }
      else if (wg >= nfull)                                  // later-round image G + e goes to helper G - 1 - (e % nh)
        for (int e = G - 1 - wg; e < p.B - G; e += nh) run(G + e);
      for (int b = wg; b < p.B; b += G) ip_wait<false>(p.ip_dcnt + (b >> 5) * IP_CNT_STRIDE, IP_TARGET, p.ip_err);
    }
  }

  f4 gE[NK], gO[NK];
#pragma unroll
  for (int dg = 0; dg < NK; ++dg) { gE[dg] = f4{0.f, 0.f, 0.f, 0.f}; gO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }

  if (n_items > 0) {
    int aoff[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) aoff[ks] = j * ROWB + (((4 * NK * w + 4 * ks + g) ^ j) << 4);
    unsigned soff[KDMA];
    mb_source_offsets<NCH, MB2_NW, KDMA>(w, lane, soff);
    int poff[4], pseg[NK];
    mb_pool_offsets<NK, MB2_NW>(w, j, g, poff, pseg);

    float sv[4];
    auto load_scores = [&](int img, int tl) {
      const int b = wg + img * G;
      const float* Srow = p.S + ((int64_t)b * QS + jq) * N;
      int t0 = tl * MB2_TT + 4 * g;
      if (n4) {
        t0 = t0 < N ? t0 : N - 4;
        const f4 v = *reinterpret_cast<const f4*>(Srow + t0);
        sv[0] = v.x; sv[1] = v.y; sv[2] = v.z; sv[3] = v.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = Srow[(t0 + r) < N ? (t0 + r) : N - 1];
      }
    };
    load_scores(0, 0);
    int pi = 0, pimg = 0, ptile = 0, pslot = 0;
    const char* psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, wg));
    auto produce = [&]() {
      if (pi < n_items) {
        const int left = N - ptile * MB2_TT;
        const unsigned limit = (unsigned)((left < MB2_TT ? left : MB2_TT) * ROWB - 16);
        mb_dma_tile<MB2_NW, KDMA>(psrc, limit, ring + pslot * SLOT, w, soff);
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++ptile == tiles_per_img) {
          ptile = 0; ++pimg;
          psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
        } else {
          psrc += SLOT;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    u4 bq[NK][3];
    float mL_j = 0.f, il_j = 0.f, dl_j = 0.f;
    int cimg = 0, ctile = 0, cslot = 0;
    // a dP row chunk: plain load, or -- rows produced inside this launch by the in-pass tasks -- an sc1 buffer load
    const __amdgpu_buffer_rsrc_t rP = ip_rsrc(p.dP, (size_t)p.B * QS * D * sizeof(float));
    auto ld_dp = [&](const float* src, int off) -> f4 {
      if constexpr (IPOK) {
        if (p.ip_dy) return ip_load16_coherent(rP, (unsigned)((src - p.dP + off) * (int64_t)sizeof(float)));
      }
      return *reinterpret_cast<const f4*>(src + off);
    };
    if constexpr (NS >= 3 && EP_MB2_PIPE) {
      // software-pipelined like the forward (see there): dA MFMAs of tile it, then the pooling MFMAs of tile it-1 between
      // the two barriers; saved scores of tile it+1 and the ring refill (tile it+2) behind the second one, in that order
      // (the plain loads first: the counted wait at the top then covers them)
      float pwgt[4] = {0.f, 0.f, 0.f, 0.f};
      const char* ptile_ = ring;
      for (int it = 0; it <= n_items; ++it) {
        const bool live = it < n_items;
        if (live) mb_wait_vmcnt((pi - 1 - it) * KDMA);
        mb_barrier();
        const int n0 = ctile * MB2_TT;
        const int nvalid = (N - n0) < MB2_TT ? (N - n0) : MB2_TT;
        const char* tile = ring + cslot * SLOT;
        if (live && ctile == 0) {
          const int b = wg + cimg * G;                 // new image: its dP rows (this wave's slice) and ML row; the other
          const float* src = p.dP + ((int64_t)b * QS + jq) * D + 32 * NK * w + 8 * g;      // workgroup of the CU covers the wait
          const f4 hml = *reinterpret_cast<const f4*>(p.ML + ((int64_t)b * QS + jq) * 4);
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const f4 a = ld_dp(src, 32 * ks), c = ld_dp(src, 32 * ks + 4);
            float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            if (j >= Q) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            mb_split8(v, bq[ks]);
          }
          mL_j = hml.x * MB_LOG2E; il_j = 1.0f / hml.y; dl_j = hml.z;
          if (p.dyv) {
            const int Dq = p.Dv / Q;
            const float* dyr = p.dyv + (int64_t)b * p.Dv + jq * Dq;
            const float* yr = p.yv + (int64_t)b * p.Dv + jq * Dq;
            float acc = 0.f;
            for (int c = 4 * g; c < Dq; c += 16) {
              const f4 a = *reinterpret_cast<const f4*>(dyr + c), y4 = *reinterpret_cast<const f4*>(yr + c);
              acc = fmaf(a.x, y4.x, acc); acc = fmaf(a.y, y4.y, acc); acc = fmaf(a.z, y4.z, acc); acc = fmaf(a.w, y4.w, acc);
            }
            dl_j = mb_q4_sum(acc);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // keeps the counted waits below exact
        }
        if (live) mb2_scores<NK, NT>(tile, aoff, bq, spart, w, lane, [] {});      // dA partial blocks
        if (it > 0) mb2_pool<NK, NT>(ptile_, poff, pseg, pwgt, gE, gO);
        if (!live) break;
        mb_barrier();                                  // partial blocks complete; the slot of tile it-1 is free
        float cur[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[e] = sv[e];
        {
          int nimg = cimg, ntile = ctile + 1;
          if (ntile == tiles_per_img) { ntile = 0; ++nimg; }
          if (nimg < n_img) load_scores(nimg, ntile);
        }
        produce();
        float u[4];
        mb2_gather(spart, lane, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = __builtin_amdgcn_exp2f(fmaf(cur[e], MB_LOG2E, -mL_j)) * il_j;
          pwgt[e] = ((4 * g + e) < nvalid && j < Q) ? a * (u[e] - dl_j) : 0.f;
        }
        ptile_ = tile;
        cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
        if (++ctile == tiles_per_img) { ctile = 0; ++cimg; }
      }
    } else
    for (int it = 0; it < n_items; ++it) {
      mb_wait_vmcnt((pi - 1 - it) * KDMA);
      mb_barrier();
      const int n0 = ctile * MB2_TT;
      const int nvalid = (N - n0) < MB2_TT ? (N - n0) : MB2_TT;
      const char* tile = ring + cslot * SLOT;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if (ctile == 0) {                              // new image: its dP rows (this wave's slice) and ML row; the other
        const int b = wg + cimg * G;                 // workgroup of the CU covers the wait
        const float* src = p.dP + ((int64_t)b * QS + jq) * D + 32 * NK * w + 8 * g;
        const f4 hml = *reinterpret_cast<const f4*>(p.ML + ((int64_t)b * QS + jq) * 4);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const f4 a = ld_dp(src, 32 * ks), c = ld_dp(src, 32 * ks + 4);
          float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
          if (j >= Q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
          }
          mb_split8(v, bq[ks]);
        }
        mL_j = hml.x * MB_LOG2E; il_j = 1.0f / hml.y; dl_j = hml.z;
        if (p.dyv) {
          // the softmax-correction term delta[b, j] = dy[b, j-slice] . y[b, j-slice] (= dP[b,j] . P[b,j]) computed here instead
          // of being read from ML[b,j,2] (no ep_delta_kernel launch in front of the pass): lane (j, g) sums quarter g of the
          // slice in float4 steps, the four lanes of a query combine in fixed order -- every wave gets the same bits
          const int Dq = p.Dv / Q;
          const float* dyr = p.dyv + (int64_t)b * p.Dv + jq * Dq;
          const float* yr = p.yv + (int64_t)b * p.Dv + jq * Dq;
          float acc = 0.f;
          for (int c = 4 * g; c < Dq; c += 16) {
            const f4 a = *reinterpret_cast<const f4*>(dyr + c), y4 = *reinterpret_cast<const f4*>(yr + c);
            acc = fmaf(a.x, y4.x, acc); acc = fmaf(a.y, y4.y, acc); acc = fmaf(a.z, y4.z, acc); acc = fmaf(a.w, y4.w, acc);
          }
          dl_j = mb_q4_sum(acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // keeps the counted waits below exact
      }
      float cur[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[e] = sv[e];
      {
        int nimg = cimg, ntile = ctile + 1;
        if (ntile == tiles_per_img) { ntile = 0; ++nimg; }
        if (nimg < n_img) load_scores(nimg, ntile);
      }
      mb2_scores<NK, NT>(tile, aoff, bq, spart, w, lane, produce);
      mb_barrier();
      float u[4], wgt[4];
      mb2_gather(spart, lane, u);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = __builtin_amdgcn_exp2f(fmaf(cur[e], MB_LOG2E, -mL_j)) * il_j;
        wgt[e] = ((4 * g + e) < nvalid && j < Q) ? a * (u[e] - dl_j) : 0.f;
      }
      mb2_pool<NK, NT>(tile, poff, pseg, wgt, gE, gO);
      if (++ctile == tiles_per_img) { ctile = 0; ++cimg; }
    }
  }
  if (j < Q) {
    float* Gq = p.Gpart + ((int64_t)wg * Q + j) * D + 32 * NK * w + 8 * g;
#pragma unroll
    for (int dg = 0; dg < NK; ++dg) mb_store8(Gq + 32 * dg, gE[dg], gO[dg], 1.0f);
  }
}

// =======================================================================================
// 17 .. 32 queries in ONE read of the bf16 tokens (round 5): the reference's default --ep_queries 32 (reference
// main_linprobe.py:113) used to run as two 16-query launches of the form above.  Here one 8-wave workgroup per CU shares a ring of
// 32-token tiles: wave w = (qb, kq) owns query block qb = w >> 2 (queries 16 qb .. 16 qb + 15) and D-quarter kq = w & 3 for both
// contractions -- the arithmetic of the 8-wave form at the top of this file (two 16-token MFMA blocks per tile, the pooling
// contraction on v_mfma_f32_16x16x32_bf16 with K = the 32 tokens of the tile) with the waves split over query blocks instead of
// all waves holding all queries: the cross-wave score exchange is 4 records per block, the softmax / operand-split work per
// wave covers ITS block only.  (A first version with 16-token tiles ran 165 / 159 us per pass at 1024 x 256 x 768: its ring alone,
// two barriers per 24 KiB tile on one workgroup per CU, moved 4.4 TB/s; 32-token tiles halve the rendezvous per byte.)
// D = 128 k (k = 2, 3, 4, 6).
// =======================================================================================
int mb2_launch_amp(int NK, bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side, size_t lds);
#ifndef EP_MB_AMP_TU
constexpr int MBQ_NW = 8, MBQ_KQ = 4;
#ifndef EP_MBQ_ABLATE
#define EP_MBQ_ABLATE 0               // diagnostic builds of the forward: 1 ring + barriers only, 2 no pooling, 4 no score MFMAs; results are wrong
#endif

template <int NK>                     // D = 128 * NK: a wave's D-quarter is 32 * NK channels (NK k-steps of the score MFMAs)
struct MbqCfg {
  static constexpr int D = 128 * NK;
  static constexpr int ROWB = 2 * D;
  static constexpr int NCH = D / 8;
  static constexpr int SLOT = MB_TT * ROWB;
  static constexpr int KDMA = SLOT / (MBQ_NW * 1024);   // = NK
  static constexpr int SPART = 2 * MBQ_NW * 1024;       // partial score records [query block][token block][wave of the block][lane] f4
  static constexpr int LDS_TOTAL = 160 * 1024;
  static constexpr int nslot() {
    int ns = (LDS_TOTAL - SPART) / SLOT;
    ns = ns > 6 ? 6 : ns;
    while (ns > 3 && (ns - 2) * KDMA > 16) --ns;        // (mb_wait_vmcnt counts up to 16)
    return ns;
  }
  static constexpr int NSLOT = nslot();
  static constexpr int LDS = NSLOT * SLOT + SPART;
  static_assert(SLOT % (MBQ_NW * 1024) == 0 && NSLOT >= 3, "the two-block form needs D = 128 k and a ring of three tiles");
};

template <int NK>
__global__ __launch_bounds__(MBQ_NW * 64, 1) void ep_pool_mbq_fwd_kernel(PoolParams p) {
  using C = MbqCfg<NK>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT, KDMA = C::KDMA, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int qb = w >> 2, kq = w & 3;
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + MB_TT - 1) / MB_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  const int j = lane & 15, g = lane >> 4;
  const int qj = 16 * qb + j;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x);
  char* sblk = spart + qb * (2 * MBQ_KQ * 1024);     // the 2 x 4 records of this wave's query block

  // B operand of the score MFMAs: query qj over the wave's D-quarter, pre-scaled like the reference (ep.py:39), three bf16 terms
  u4 bq[NK][3];
  int aoff[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (qj < Q) {
      const float* src = p.cls + (int64_t)qj * D + 32 * NK * kq + 32 * ks + 8 * g;
      const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
      v[0] = a.x * p.scale; v[1] = a.y * p.scale; v[2] = a.z * p.scale; v[3] = a.w * p.scale;
      v[4] = b.x * p.scale; v[5] = b.y * p.scale; v[6] = b.z * p.scale; v[7] = b.w * p.scale;
    }
    mb_split8(v, bq[ks]);
    aoff[ks] = j * ROWB + (((4 * NK * kq + 4 * ks + g) ^ j) << 4);
  }
  unsigned soff[KDMA];
  mb_source_offsets<NCH, MBQ_NW, KDMA>(w, lane, soff);
  int poff[4], pseg[NK];
  mb_pool_offsets<NK, MBQ_KQ>(kq, j, g, poff, pseg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, wg));
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * MB_TT;
      const unsigned limit = (unsigned)((left < MB_TT ? left : MB_TT) * ROWB - 16);
      mb_dma_tile<MBQ_NW, KDMA>(psrc, limit, ring + pslot * SLOT, w, soff);
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
      } else {
        psrc += SLOT;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 accE[NK], accO[NK];
  float m_j = -INFINITY, mL_j = -INFINITY, lsum = 0.f;     // per lane: running max / partial sum of query qj
  int cslot = 0, it = 0;
  for (int img = 0; img < n_img; ++img) {
    const int b = wg + img * G;
    m_j = -INFINITY; mL_j = -INFINITY; lsum = 0.f;
#pragma unroll
    for (int dg = 0; dg < NK; ++dg) { accE[dg] = f4{0.f, 0.f, 0.f, 0.f}; accO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }
    for (int t = 0; t < tiles_per_img; ++t, ++it) {
      mb_wait_vmcnt((pi - 1 - it) * KDMA);
      mb_barrier();                                  // tile `it` landed everywhere; the slot of tile it-1 is free
      const int n0 = t * MB_TT;
      const int nvalid = (N - n0) < MB_TT ? (N - n0) : MB_TT;
      const char* tile = ring + cslot * SLOT;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if constexpr (EP_MBQ_ABLATE == 1) { produce(); mb_barrier(); continue; }
      if constexpr (EP_MBQ_ABLATE == 4) { produce(); } else
      mb_scores<NK, MBQ_KQ, false>(tile, aoff, bq, sblk, kq, lane, produce);
      mb_barrier();                                  // all partial score blocks are in the scratch
      float sc[8], ue[8];
      mb_gather<MBQ_KQ, false>(sblk, lane, sc);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ue[e] = (16 * (e >> 2) + 4 * g + (e & 3)) < nvalid ? sc[e] : -INFINITY;
        mx = fmaxf(mx, ue[e]);
      }
      if (__builtin_amdgcn_ballot_w64(mx > m_j + MB_LAZY_MAX_THR) != 0ull) {    // rare
        const float mn = fmaxf(m_j, mb_q4_max(mx));
        const float f = __builtin_amdgcn_exp2f((m_j - mn) * MB_LOG2E);           // m = -inf -> 0
        m_j = mn; mL_j = mn * MB_LOG2E;
        lsum *= f;
#pragma unroll
        for (int dg = 0; dg < NK; ++dg) { accE[dg] *= f; accO[dg] *= f; }       // my column is query qj
      }
      float wgt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        wgt[e] = __builtin_amdgcn_exp2f(fmaf(ue[e], MB_LOG2E, -mL_j));          // invalid tokens: 0
        lsum += wgt[e];
      }
      // the four waves of a block hold the same scores: wave kq writes token group kq of both token blocks
      if (g == kq && qj < Q) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const int t0 = 16 * blk + 4 * g;
          float* Srow = p.S + ((int64_t)b * QS + qj) * N + n0 + t0;
          if (n4) {
            if (t0 < nvalid) *reinterpret_cast<f4*>(Srow) = f4{sc[4 * blk], sc[4 * blk + 1], sc[4 * blk + 2], sc[4 * blk + 3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t0 + r < nvalid) Srow[r] = sc[4 * blk + r];
          }
        }
      }
      if constexpr (EP_MBQ_ABLATE != 2) mb_pool<NK, MBQ_KQ>(tile, poff, pseg, wgt, accE, accO);
    }
    const float l = mb_q4_sum(lsum);
    const float inv = 1.0f / l;
    if (qj < Q) {
      float* Pq = p.P + ((int64_t)b * QS + qj) * D + 32 * NK * kq + 8 * g;
#pragma unroll
      for (int dg = 0; dg < NK; ++dg) mb_store8(Pq + 32 * dg, accE[dg], accO[dg], inv);
      if (kq == 0 && g == 0) {
        const f4 rec = {m_j, l, 0.f, 0.f};
        *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + qj) * 4) = rec;
      }
    }
  }
}

// backward: the same pair with dP in place of the queries and A*(dA-delta) in place of the softmax weights.  The dP rows of
// an image (this wave's D-quarter of its query block: read straight from global memory at the image's first tile -- the ring keeps
// streaming meanwhile) are split into bf16 terms once per image; the saved scores of a tile are fetched one tile ahead.
template <int NK>
__global__ __launch_bounds__(MBQ_NW * 64, 1) void ep_pool_mbq_bwd_kernel(PoolParams p) {
  using C = MbqCfg<NK>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT, KDMA = C::KDMA, NCH = C::NCH;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int qb = w >> 2, kq = w & 3;
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + MB_TT - 1) / MB_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  const int j = lane & 15, g = lane >> 4;
  const int qj = 16 * qb + j;
  const bool live = qj < Q;
  const int jq = live ? qj : Q - 1;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x);
  char* sblk = spart + qb * (2 * MBQ_KQ * 1024);

  f4 gE[NK], gO[NK];
#pragma unroll
  for (int dg = 0; dg < NK; ++dg) { gE[dg] = f4{0.f, 0.f, 0.f, 0.f}; gO[dg] = f4{0.f, 0.f, 0.f, 0.f}; }

  if (n_items > 0) {
    int aoff[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) aoff[ks] = j * ROWB + (((4 * NK * kq + 4 * ks + g) ^ j) << 4);
    unsigned soff[KDMA];
    mb_source_offsets<NCH, MBQ_NW, KDMA>(w, lane, soff);
    int poff[4], pseg[NK];
    mb_pool_offsets<NK, MBQ_KQ>(kq, j, g, poff, pseg);

    // saved scores of tile (img, tile): S[b, qj, n0 + 16 blk + 4 g + r]
    float sv[8];
    auto load_scores = [&](int img, int tl) {
      const int b = wg + img * G;
      const float* Srow = p.S + ((int64_t)b * QS + jq) * N;
      const int n0 = tl * MB_TT;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        int t0 = n0 + 16 * blk + 4 * g;
        if (n4) {
          t0 = t0 < N ? t0 : N - 4;
          const f4 v = *reinterpret_cast<const f4*>(Srow + t0);
          sv[4 * blk] = v.x; sv[4 * blk + 1] = v.y; sv[4 * blk + 2] = v.z; sv[4 * blk + 3] = v.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[4 * blk + r] = Srow[(t0 + r) < N ? (t0 + r) : N - 1];
        }
      }
    };
    load_scores(0, 0);
    int pi = 0, pimg = 0, ptile = 0, pslot = 0;
    const char* psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, wg));
    auto produce = [&]() {
      if (pi < n_items) {
        const int left = N - ptile * MB_TT;
        const unsigned limit = (unsigned)((left < MB_TT ? left : MB_TT) * ROWB - 16);
        mb_dma_tile<MBQ_NW, KDMA>(psrc, limit, ring + pslot * SLOT, w, soff);
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++ptile == tiles_per_img) {
          ptile = 0; ++pimg;
          psrc = reinterpret_cast<const char*>(xb + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
        } else {
          psrc += SLOT;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    u4 bq[NK][3];
    float mL_j = 0.f, il_j = 0.f, dl_j = 0.f;
    int cslot = 0, it = 0;
    for (int img = 0; img < n_img; ++img) {
      const int b = wg + img * G;
      for (int t = 0; t < tiles_per_img; ++t, ++it) {
        mb_wait_vmcnt((pi - 1 - it) * KDMA);
        mb_barrier();
        const int n0 = t * MB_TT;
        const int nvalid = (N - n0) < MB_TT ? (N - n0) : MB_TT;
        const char* tile = ring + cslot * SLOT;
        cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
        if (t == 0) {                                // new image: its dP rows (this wave's slice) and ML row
          const float* src = p.dP + ((int64_t)b * QS + jq) * D + 32 * NK * kq + 8 * g;
          const f4 hml = *reinterpret_cast<const f4*>(p.ML + ((int64_t)b * QS + jq) * 4);
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const f4 a = *reinterpret_cast<const f4*>(src + 32 * ks), c = *reinterpret_cast<const f4*>(src + 32 * ks + 4);
            float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            if (!live) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            mb_split8(v, bq[ks]);
          }
          mL_j = hml.x * MB_LOG2E; il_j = 1.0f / hml.y; dl_j = hml.z;
          if (p.dyv) {                               // delta[b, qj] = dy[b, qj-slice] . y[b, qj-slice] (see ep_pool_mb2_bwd_kernel)
            const int Dq = p.Dv / Q;
            const float* dyr = p.dyv + (int64_t)b * p.Dv + jq * Dq;
            const float* yr = p.yv + (int64_t)b * p.Dv + jq * Dq;
            float acc = 0.f;
            for (int c = 4 * g; c < Dq; c += 16) {
              const f4 a = *reinterpret_cast<const f4*>(dyr + c), y4 = *reinterpret_cast<const f4*>(yr + c);
              acc = fmaf(a.x, y4.x, acc); acc = fmaf(a.y, y4.y, acc); acc = fmaf(a.z, y4.z, acc); acc = fmaf(a.w, y4.w, acc);
            }
            dl_j = mb_q4_sum(acc);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // keeps the counted waits below exact
        }
        float cur[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cur[e] = sv[e];
        {                                            // scores of the next tile, in front of the ring refill (the counted wait covers them)
          int nimg = img, ntile = t + 1;
          if (ntile == tiles_per_img) { ntile = 0; ++nimg; }
          if (nimg < n_img) load_scores(nimg, ntile);
        }
        mb_scores<NK, MBQ_KQ, false>(tile, aoff, bq, sblk, kq, lane, produce);      // dA partial blocks
        mb_barrier();
        float u[8], wgt[8];
        mb_gather<MBQ_KQ, false>(sblk, lane, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = __builtin_amdgcn_exp2f(fmaf(cur[e], MB_LOG2E, -mL_j)) * il_j;
          wgt[e] = ((16 * (e >> 2) + 4 * g + (e & 3)) < nvalid && live) ? a * (u[e] - dl_j) : 0.f;
        }
        mb_pool<NK, MBQ_KQ>(tile, poff, pseg, wgt, gE, gO);
      }
    }
  }
  if (live) {
    float* Gq = p.Gpart + ((int64_t)wg * Q + qj) * D + 32 * NK * kq + 8 * g;
#pragma unroll
    for (int dg = 0; dg < NK; ++dg) mb_store8(Gq + 32 * dg, gE[dg], gO[dg], 1.0f);
  }
}

template <int NK>
static int mbq_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  using C = MbqCfg<NK>;
  const size_t lds = C::LDS;
  auto kf = ep_pool_mbq_fwd_kernel<NK>;
  auto kb = ep_pool_mbq_bwd_kernel<NK>;
  const void* fn = bwd ? (const void*)kb : (const void*)kf;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
  if (bwd) hipLaunchKernelGGL(kb, dim3(grid), dim3(MBQ_NW * 64), lds, st, p);
  else hipLaunchKernelGGL(kf, dim3(grid), dim3(MBQ_NW * 64), lds, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_mbq_bwd_kernel" : "ep_pool_mbq_fwd_kernel");
  return 0;
}
// bf16 tokens, shared query rows, 17 .. 32 queries: D in {256, 384, 512, 768} (D = 1024 would spill and hold a ring of two
// tiles only -- it stays on two 16-query launches of the form above)
bool mbq_supported(int D, int Q, int64_t cls_bstride) {
  return cls_bstride == 0 && Q > 16 && Q <= 32 && (D == 256 || D == 384 || D == 512 || D == 768);
}
bool mbq_takes_delta(int D, int Q, int Dv) { return mbq_supported(D, Q, 0) && Dv > 0 && Dv % (4 * Q) == 0; }
int mbq_grid(int B) { const int g = cu_count(); return g < B ? g : B; }
int mbq_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  switch (p.D) {
    case 256: return mbq_launch_one<2>(bwd, p, grid, st);
    case 384: return mbq_launch_one<3>(bwd, p, grid, st);
    case 512: return mbq_launch_one<4>(bwd, p, grid, st);
    case 768: return mbq_launch_one<6>(bwd, p, grid, st);
  }
  set_error("no 32-query bf16 matrix-core pooling kernel for D=%d", p.D);
  return EP_E_UNSUPPORTED;
}

static thread_local int* g_mb_occ_query = nullptr;

template <int NK, int NS = 3>
static int mb2_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side) {
  using C = Mb2Cfg<NK, NS>;
  size_t lds = C::LDS;
  const bool with_side = bwd && side && side->total > 0;
  SideTasks sd{};
  if (with_side) {
    sd = *side;
    sd.first_block = grid;
    if (lds < SIDE_LDS_BYTES) lds = SIDE_LDS_BYTES;
  }
  if (bwd && p.ip_dy) {
    if (!with_side || (NK != 2 && NK != 4 && NK != 6)) { set_error("in-pass dP needs the side-carrying pass at D = 256 / 512 / 768"); return EP_E_UNSUPPORTED; }
    if (lds < ip_dp_lds_bytes(NK / 2)) lds = ip_dp_lds_bytes(NK / 2);
  }
  if (g_mb_occ_query) {                              // mb_resident_blocks_per_cu(): answer instead of launching
    auto kq = ep_pool_mb2_bwd_kernel<NK, NS, true>;
    int nb = 0;
    (void)hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kq, MB2_NW * 64, lds) != hipSuccess) { (void)hipGetLastError(); nb = -1; }
    *g_mb_occ_query = nb;
    return 0;
  }
  if (p.nterms == 1) return mb2_launch_amp(NK, bwd, p, grid, st, with_side ? &sd : nullptr, lds);   // single-product instances: ep_pool_mb_amp.hip
  auto kf = ep_pool_mb2_fwd_kernel<NK, NS>;
  auto kb = ep_pool_mb2_bwd_kernel<NK, NS, false>;
  auto ks = ep_pool_mb2_bwd_kernel<NK, NS, true>;
  const void* fn = bwd ? (with_side ? (const void*)ks : (const void*)kb) : (const void*)kf;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
  if (with_side) hipLaunchKernelGGL(ks, dim3(grid + sd.total), dim3(MB2_NW * 64), lds, st, p, sd);
  else if (bwd) hipLaunchKernelGGL(kb, dim3(grid), dim3(MB2_NW * 64), lds, st, p, sd);
  else hipLaunchKernelGGL(kf, dim3(grid), dim3(MB2_NW * 64), lds, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_mb2_bwd_kernel" : "ep_pool_mb2_fwd_kernel");
  return 0;
}
// the two-workgroup form is the default for every D it supports (256, 384, 512, 768, 1024: measured faster on all of them,
// e.g. 80 / 83 us against 91 / 100 us at 256x768, 42 / 51 against 62 / 73 at 196x384); EP_POOL_MB2=0 switches it off
static bool mb2_use(int D) {
  static int v = -1;
  if (v < 0) { const char* e = getenv("EP_POOL_MB2"); v = e ? atoi(e) : 1; }
  if (v == 0) return false;
  // D = 1024: ring of two 32 KiB slots so that two workgroups fit (82 / 85 us against 93 / 101 us of the 8-wave form);
  // D = 1152 would spill (10 / 22 registers: 190 / 249 us) and stays on the 12-wave form
  return D == 256 || D == 384 || D == 512 || D == 768 || D == 1024;
}
// resident workgroups per CU of the two-workgroup form: what LDS (3 slots of 32 D bytes + 4 KiB) and registers allow
static int mb2_wgs_per_cu(int D) {
  static int ov = -1;
  if (ov < 0) { const char* e = getenv("EP_POOL_MB2_WGS"); ov = e ? atoi(e) : 0; }
  if (ov > 0) return ov;
  return D <= 384 ? 4 : 2;     // measured: 512 runs 54 us on two, 58 us on four
}
const char* mb_kernel_name(int D, bool bwd) {
  if (mb2_use(D)) return bwd ? "ep_pool_mb2_bwd_kernel" : "ep_pool_mb2_fwd_kernel";
  return bwd ? "ep_pool_mb_bwd_kernel" : "ep_pool_mb_fwd_kernel";
}
int mb_grid(int D, int B) {
  const int g = cu_count() * (mb2_use(D) ? mb2_wgs_per_cu(D) : 1);
  return g < B ? g : B;
}

// ---------------------------------------------------------------------------------------
template <int NK, int NW, bool PK>
static int mb_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  using C = MbCfg<NK, NW, PK>;
  static_assert(C::VALID, "bf16 matrix-core pooling configuration does not fit");
  const size_t lds = (size_t)C::NSLOT * C::SLOT + C::SPART;
  auto kf = ep_pool_mb_fwd_kernel<NK, NW, PK>;
  auto kb = ep_pool_mb_bwd_kernel<NK, NW, PK>;
  const void* fn = bwd ? (const void*)kb : (const void*)kf;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
  if (bwd) hipLaunchKernelGGL(kb, dim3(grid), dim3(NW * 64), lds, st, p);
  else hipLaunchKernelGGL(kf, dim3(grid), dim3(NW * 64), lds, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_mb_bwd_kernel" : "ep_pool_mb_fwd_kernel");
  return 0;
}

// bf16 tokens, shared query rows.  Up to 8 queries (packed exchange): D in {256, 384, 512, 768, 1024, 1152};
// 9-16 queries: D in {256, 512, 768, 1024}.
bool mb_supported(int D, int Q, int64_t cls_bstride) {
  if (cls_bstride != 0 || Q < 1 || Q > 16) return false;
  if (D == 256 || D == 512 || D == 768 || D == 1024) return true;
  if (D == 384) return true;
  return Q <= 8 && D == 1152;
}

static int mb_variant() {             // diagnostic: EP_POOL_MB_WAVES=12 runs D = 768 on the 12-wave form (measured 94 vs 90 us)
  static int v = -1;
  if (v < 0) { const char* e = getenv("EP_POOL_MB_WAVES"); v = e ? atoi(e) : 0; }
  return v;
}

// the two-workgroup form carries the step's side work (SideTasks) and computes the softmax-correction rows itself (dyv / yv)
bool mb_takes_side(int D) { return mb2_use(D); }
// In-pass dP (ep_inpass.h) inside the second pass of this shape: Q = 8, D = 256 KT <= 768, and the WHOLE pooling grid
// resident at once (workgroups wait on row blocks that other workgroups of the launch produce) -- asked of the runtime for
// the side-carrying kernel with the tasks' LDS.  Cached per D.
bool mb_takes_inpass_dp(const PoolParams& p) {
  if (!mb2_use(p.D) || p.Q != 8 || (p.D != 256 && p.D != 512 && p.D != 768)) return false;
  static int res[4] = {0, 0, 0, 0};                  // per D / 256: 0 unknown, else blocks per CU + 1 (INT_MAX: cannot tell)
  int& r = res[p.D / 256];
  if (r == 0) {
    PoolParams q = p;
    int dummy = 0, nb = -1;
    q.ip_dy = reinterpret_cast<const float*>(&dummy);
    SideTasks sd{};
    sd.total = 1;
    g_mb_occ_query = &nb;
    const int rc = mb_launch(true, q, 1, nullptr, &sd);
    g_mb_occ_query = nullptr;
    r = (rc != 0 || nb < 0) ? 0x7fffffff : nb + 1;
  }
  return r == 0x7fffffff || (int64_t)(r - 1) * cu_count() >= mb_grid(p.D, p.B);
}
bool mb_takes_delta(int D, int Q, int Dv) { return mb2_use(D) && Dv > 0 && Dv % (4 * Q) == 0; }

int mb_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side) {
  const bool pk = p.Q <= 8;
  if (mb2_use(p.D)) {
    switch (p.D) {
      case 256: return mb2_launch_one<2>(bwd, p, grid, st, side);
      case 384: return mb2_launch_one<3>(bwd, p, grid, st, side);
      case 512: return mb2_launch_one<4>(bwd, p, grid, st, side);
      case 768: return mb2_launch_one<6>(bwd, p, grid, st, side);
      case 1024: return mb2_launch_one<8, 2>(bwd, p, grid, st, side);
    }
  }
  if (bwd && side && side->total > 0) { set_error("side tasks need the two-workgroup bf16 matrix-core pass"); return EP_E_UNSUPPORTED; }
  switch (p.D) {
    case 256: return pk ? mb_launch_one<1, 8, true>(bwd, p, grid, st) : mb_launch_one<1, 8, false>(bwd, p, grid, st);
    case 384: if (pk) return mb_launch_one<1, 12, true>(bwd, p, grid, st); break;
    case 512: return pk ? mb_launch_one<2, 8, true>(bwd, p, grid, st) : mb_launch_one<2, 8, false>(bwd, p, grid, st);
    case 768:
      if (!pk) return mb_launch_one<3, 8, false>(bwd, p, grid, st);
      return mb_variant() == 12 ? mb_launch_one<2, 12, true>(bwd, p, grid, st) : mb_launch_one<3, 8, true>(bwd, p, grid, st);
    case 1024: return pk ? mb_launch_one<4, 8, true>(bwd, p, grid, st) : mb_launch_one<4, 8, false>(bwd, p, grid, st);
    case 1152: if (pk) return mb_launch_one<3, 12, true>(bwd, p, grid, st); break;
  }
  set_error("no bf16 matrix-core pooling kernel for D=%d, Q=%d", p.D, p.Q);
  return EP_E_UNSUPPORTED;
}

#else   // EP_MB_AMP_TU: the single-product (AMP-bf16, NT = 1) instances of the mb2 kernels, a translation unit of their own
          // (ep_pool_mb_amp.hip) so that they compile beside this file instead of behind it
template <int NK, int NS>
static int mb2_launch_amp_one(bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side, size_t lds) {
  if (p.ip_dy) { set_error("the single-product (AMP-bf16) bf16-token passes carry no in-pass dP"); return EP_E_UNSUPPORTED; }
  const bool with_side = bwd && side && side->total > 0;
  SideTasks sd{};
  if (with_side) sd = *side;                          // (first_block set by the caller)
  auto kf1 = ep_pool_mb2_fwd_kernel<NK, NS, 1>;
  auto kb1 = ep_pool_mb2_bwd_kernel<NK, NS, false, 1>;
  auto ks1 = ep_pool_mb2_bwd_kernel<NK, NS, true, 1>;
  const void* fn1 = bwd ? (with_side ? (const void*)ks1 : (const void*)kb1) : (const void*)kf1;
  hipError_t e1 = hipFuncSetAttribute(fn1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e1 != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e1)); return (int)e1; }
  if (with_side) hipLaunchKernelGGL(ks1, dim3(grid + sd.total), dim3(MB2_NW * 64), lds, st, p, sd);
  else if (bwd) hipLaunchKernelGGL(kb1, dim3(grid), dim3(MB2_NW * 64), lds, st, p, sd);
  else hipLaunchKernelGGL(kf1, dim3(grid), dim3(MB2_NW * 64), lds, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_mb2_bwd_kernel (single product)" : "ep_pool_mb2_fwd_kernel (single product)");
  return 0;
}
int mb2_launch_amp(int NK, bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side, size_t lds) {
  switch (NK) {
    case 2: return mb2_launch_amp_one<2, 3>(bwd, p, grid, st, side, lds);
    case 3: return mb2_launch_amp_one<3, 3>(bwd, p, grid, st, side, lds);
    case 4: return mb2_launch_amp_one<4, 3>(bwd, p, grid, st, side, lds);
    case 6: return mb2_launch_amp_one<6, 3>(bwd, p, grid, st, side, lds);
    case 8: return mb2_launch_amp_one<8, 2>(bwd, p, grid, st, side, lds);
  }
  set_error("mb2_launch_amp: no instance for NK = %d", NK);
  return EP_E_UNSUPPORTED;
}
#endif  // EP_MB_AMP_TU

}  // namespace ep
