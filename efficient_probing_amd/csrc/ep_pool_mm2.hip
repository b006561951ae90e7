// EP attentive pooling on fp32 tokens with 17 .. 32 queries in ONE read of the tokens (gfx950 / CDNA4; round 5).
//
// The reference's default and every published row train with --ep_queries 32 (reference main_linprobe.py:113,
// poolings/ep.py:35-44).  Up to round 4 that ran as two 16-query chunks of ep_pool_mm.hip (forward) and as one pass of the
// vector-ALU kernel (backward): two token reads, or 4 k FMA per token on the vector pipe.  Here one 8-wave workgroup per CU
// keeps the ring tile resident and runs BOTH 16-query blocks against it, on v_mfma_f32_16x16x4_f32 (exact fp32):
//
//   scores   S[t][q]   = sum_d x[t][d] * (cls[q][d]*scale)        A = x tile,    B = queries
//   pooling  P^T[d][q] = sum_t x[t][d] * softmax-weight[q][t]     A = x tile^T,  B = weights
//
// Wave w = (qb, kq): query block qb = w >> 2 (queries 16 qb .. 16 qb + 15) and D-quarter kq = w & 3, for both contractions.
// The four waves of a query block sum their partial score blocks through LDS (1 KiB records; every wave reads the same lane
// slot of the four records: 4 b128 reads), each runs the online softmax of its block (4 values per lane) and pools its
// D-quarter for its 16 queries.  Against the first form of this kernel (every wave a D-eighth for all 32 queries: 16 b128
// gather reads and 8 exponentials per lane and tile, redundantly on 8 waves) the vector-ALU / LDS work between the two MFMA
// phases of a tile is a quarter; the MFMA count per wave is the same (D/8 per tile), each A operand is read from LDS by the
// two waves that share a SIMD (w and w + 4).
// (Two alternative tile schedules -- staggered halves, pipelined pooling -- are kept behind EP_MM2_STAGGER: measured equal.)
// The pass is matrix-pipe bound: 2 * 2 * N * D * 32 FLOP per image, 25.8 GFLOP at 1024 x 256 x 768 = 164 us at the 157 TFLOP/s
// fp32-matrix peak -- within 1.3 x of the HBM time of the fp32 tokens (805 MB at 6.3 TB/s = 128 us).
// The backward is the same pair with dP in place of the queries (two header items per image) and A*(dA-delta) in place of
// the softmax weights.
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"

namespace ep {

typedef __attribute__((address_space(3))) void* m2_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* m2_gptr_t;

#ifndef EP_MM2_ABLATE
#define EP_MM2_ABLATE 0               // diagnostic builds of the forward (tools/build_variant_one.sh): 1 ring + barriers only, 4 all MFMAs
#endif                                // but no gather / softmax, 5 = 4 without ring refills; results are wrong
#ifndef EP_MM2_STAGGER
#define EP_MM2_STAGGER 0              // tile schedule: 0 (shipped) every wave finishes a tile inside its iteration; 1 waves 4-7 carry the
#endif                                // pooling MFMAs of a tile into the next iteration (stagger); 2 ALL waves do, and issue them right behind
                                      // the score MFMAs of the next tile: one matrix and one vector phase per tile (pipe).  Measured EQUAL
                                      // at 256 x 768 (fwd / bwd alone: 301 / 272, 305 / 270, 307 / 278 us) and at D = 384 / 512: the exact-fp32
                                      // MFMA runs at the fp32 VECTOR rate and, as measured here, does not overlap the SIMD partner's vector
                                      // instructions, so re-ordering phases between the two waves of a SIMD is zero-sum (EXPERIMENTS.md r5)
#ifndef EP_MM2_PRIO
#define EP_MM2_PRIO 0                 // static s_setprio of waves 4-7, the younger wave of every SIMD (A/B builds; measured zero-sum)
#endif
#ifndef EP_MM2_CLK
#define EP_MM2_CLK 0                  // diagnostic builds: 1 = shader-clock / 100 MHz real-time deltas of the forward pass (the clock
#endif                                // the chip holds under this kernel) and per-phase cycle counts of workgroup 0, printed by the launcher
constexpr int M2_TT = 16;             // tokens per tile
constexpr int M2_NW = 8;              // waves per workgroup (one workgroup per CU)
constexpr int M2_KQ = 4;              // D-quarters (waves per query block)
constexpr float M2_LOG2E = 1.4426950408889634f;
constexpr float M2_LAZY_MAX_THR = 12.0f;

template <int NG>
struct Mm2Cfg {
  static constexpr int D = 128 * NG;
  static constexpr int NB = 2 * NG;                            // 16-channel blocks of a D-quarter
  static constexpr int ROWB = 4 * D;
  static constexpr int SLOT = M2_TT * ROWB;
  static constexpr int KDMA = NG;                              // 1 KiB DMA pieces per wave per tile
  static constexpr int SPART = M2_NW * 1024;                   // partial score blocks [wave][lane] f4
  static constexpr int SMALL = 2560;                           // per slot (backward): S tiles of the two blocks (2 x 1 KiB) | ML rows of a block (256 B) | pad
  static constexpr int LDS_TOTAL = 160 * 1024;
  static constexpr int nslot(bool bwd) {
    int ns = (LDS_TOTAL - SPART) / (SLOT + (bwd ? SMALL : 0));
    return ns > 4 ? 4 : ns;
  }
  static constexpr size_t lds_bytes(bool bwd) { return (size_t)nslot(bwd) * (SLOT + (bwd ? SMALL : 0)) + SPART; }
};

template <int N>
__device__ __forceinline__ void m2_wait_vmcnt_imm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate out of range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void m2_wait_vmcnt(int n) {
#define EP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    EP_W(0) EP_W(1) EP_W(2) EP_W(3) EP_W(4) EP_W(5) EP_W(6) EP_W(7) EP_W(8) EP_W(9)
    EP_W(10) EP_W(11) EP_W(12) EP_W(13) EP_W(14) EP_W(15) EP_W(16) EP_W(17) EP_W(18) EP_W(19)
    EP_W(20) EP_W(21) EP_W(22) EP_W(23) EP_W(24) EP_W(25) EP_W(26) EP_W(27) EP_W(28) EP_W(29) EP_W(30)
    default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
  }
#undef EP_W
}
__device__ __forceinline__ void m2_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// combine a per-lane value over the 4 lanes that share a query (lane, lane^16, lane^32, lane^48)
__device__ __forceinline__ float m2_q4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float m2_q4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// LDS position (t, c') of a tile holds source chunk c' ^ (t & 15) of row t (same swizzle as ep_pool_mm.hip)
template <int NG>
__device__ __forceinline__ void m2_source_offsets(int w, int lane, unsigned (&soff)[NG]) {
  constexpr int NCHUNK = 32 * NG;
#pragma unroll
  for (int jj = 0; jj < NG; ++jj) {
    const int pos = (w + M2_NW * jj) * 64 + lane;
    const int t = pos / NCHUNK, c = pos - t * NCHUNK;
    soff[jj] = (unsigned)(t * (512 * NG) + ((c ^ (t & 15)) << 4));
  }
}
template <int NG>
__device__ __forceinline__ void m2_dma_tile(const char* src, unsigned limit, char* slot, int w, const unsigned (&soff)[NG]) {
#pragma unroll
  for (int jj = 0; jj < NG; ++jj) {
    const unsigned off = soff[jj] < limit ? soff[jj] : limit;
    __builtin_amdgcn_global_load_lds((m2_gptr_t)(src + off), (m2_lds_ptr_t)(slot + (w + M2_NW * jj) * 1024), 16, 0, EP_DMA_AUX);
  }
}

// step 1: 16 tokens x 16 queries over this wave's D-quarter -> its record of the LDS scratch.  Two accumulators (alternating
// k-steps) keep the dependent chain off the critical path when the SIMD partner is not issuing; `mid` (the ring refill)
// goes behind the first MFMAs.  A operand: lane (i = token, kk): chunk 4*(NB*kq + g) + kk of row i.  The operands are read
// in two halves (the second into the first's registers).
// (`aoff(g)`: byte offset of operand g -- a table in registers, or recomputed per tile where the registers are short: D = 1152)
template <int NB, typename AO, typename F>
__device__ __forceinline__ void m2_scores(const char* tile, AO&& aoff, const float (&bq)[NB][4], char* spart, int w, int lane,
                                          F&& mid) {
  f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int H = NB / 2;
  f4 xa[H];
#pragma unroll
  for (int g = 0; g < H; ++g) xa[g] = *reinterpret_cast<const f4*>(tile + aoff(g));
#pragma unroll
  for (int g = 0; g < H; ++g) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].x, bq[g][0], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].y, bq[g][1], a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].z, bq[g][2], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].w, bq[g][3], a1, 0, 0, 0);
    if (g == 0) mid();
  }
  f4 xb[NB - H];
#pragma unroll
  for (int g = 0; g < NB - H; ++g) xb[g] = *reinterpret_cast<const f4*>(tile + aoff(H + g));
#pragma unroll
  for (int g = 0; g < NB - H; ++g) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].x, bq[H + g][0], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].y, bq[H + g][1], a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].z, bq[H + g][2], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].w, bq[H + g][3], a1, 0, 0, 0);
  }
  *reinterpret_cast<f4*>(spart + (w * 64 + lane) * 16) = a0 + a1;
}
// full scores of (query 16*qb + j, tokens 4*kk + r, r = 0..3): the MFMA D layout of a score block (col = query, row = token: lane
// 16*rowgroup + col holds rows 4*rowgroup + r) is read back at the SAME lane slot of the records of the block's four waves --
// 4 b128 reads, summed in wave order (every wave of the block gets the same bits).
__device__ __forceinline__ f4 m2_gather(const char* spart, int qb, int lane) {
  const char* base = spart + (qb * M2_KQ * 64 + lane) * 16;
  f4 v = *reinterpret_cast<const f4*>(base);
#pragma unroll
  for (int ws = 1; ws < M2_KQ; ++ws) v += *reinterpret_cast<const f4*>(base + ws * 1024);
  return v;
}
// The pooling MFMAs want their B operand as (k-step s of lane (j, kk)) <-> token 4*s + kk -- with that order the A-operand
// reads below need one per-lane address and immediates -- while the softmax arithmetic leaves register r of lane (j, kk) <-> token
// 4*kk + r: a 4 x 4 transpose between the four 16-lane rows of the wave and the four registers, two v_permlane32_swap and two
// v_permlane16_swap (gfx950).
__device__ __forceinline__ f4 m2_transpose4(const f4& v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.z), false, false);   // rows {0,1} <-> {2,3}
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.y), __float_as_uint(v.w), false, false);
  auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);                                   // rows {0,2} <-> {1,3}
  auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  return f4{__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(d[0]), __uint_as_float(d[1])};
}
// step 3, operands: the A operand of pooling k-step s is x[token 4*s + kk][channel of row i] for lane (i, kk).
//
// D = 256 k (NB % 4 == 0): the 16-row output blocks take their channels INTERLEAVED -- row i of block 4u + m <-> channel
// 64 u + 4 i + m of the wave's D-quarter -- so that one ds_read_b128 (4 consecutive channels of one token) feeds the same k-step
// of four blocks: NB b128 reads per tile instead of 4 NB b32 reads with 2-way bank conflicts (measured: the b32 form kept the LDS
// busy for 1.5 k of a tile's ~9.7 k cycles and the pooling MFMAs waiting on it).  Chunk c = 4*NB*kq + 16*u + i of row t = 4*s + kk
// sits at chunk position c ^ (t & 15): byte offset = pb[s] (per lane: t*ROWB + 64*NB*kq + 16*(i ^ (4*s + kk))) + 256*u; the
// sixteen lanes of a row read one whole 256-byte group: conflict-free.  The pooled rows come out in the same interleaved
// order and are stored as float4 across the four blocks of a group (m2_store_rows).
// Other D (NB % 4 != 0): blocks of 16 consecutive channels, b32 reads: chunk c = 4*cb + (i >> 2), cb = NB*kq + blk, at position
// 4*(cb with its low 2 bits ^ s) + ((i >> 2) ^ kk): byte offset = [kk*ROWB + 16*((i>>2)^kk) + 4*(i&3)] (`plane`) + 4*s*ROWB +
// 64*swz(cb, s).
template <int NB>
struct M2Pool {
  static constexpr bool WIDE = NB % 4 == 0;
  int pb[WIDE ? 4 : 1];
};
template <int NB>
__device__ __forceinline__ void m2_pool_offsets(int kq, int j, int kk, M2Pool<NB>& o) {
  constexpr int ROWB = 256 * NB;
  if constexpr (M2Pool<NB>::WIDE) {
#pragma unroll
    for (int s = 0; s < 4; ++s) o.pb[s] = (4 * s + kk) * ROWB + 64 * NB * kq + 16 * (j ^ (4 * s + kk));
  } else {
    o.pb[0] = kk * ROWB + 16 * ((j >> 2) ^ kk) + 4 * (j & 3);
  }
}
template <int NB>
__device__ __forceinline__ void m2_pool_load(const char* tile, const M2Pool<NB>& o, int kq, float (&xa)[4][NB]) {
  constexpr int ROWB = 256 * NB;
  if constexpr (M2Pool<NB>::WIDE) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int u = 0; u < NB / 4; ++u) {
        const f4 v = *reinterpret_cast<const f4*>(tile + o.pb[s] + 256 * u);
        xa[s][4 * u] = v.x; xa[s][4 * u + 1] = v.y; xa[s][4 * u + 2] = v.z; xa[s][4 * u + 3] = v.w;
      }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const int cb = NB * kq + blk;
        const int uni = 4 * s * ROWB + 64 * ((cb & ~3) | ((cb & 3) ^ s));
        xa[s][blk] = *reinterpret_cast<const float*>(tile + uni + o.pb[0]);
      }
  }
}
// rows 4*kk + r of the NB output blocks of lane (column j, row group kk) -> dst[channel] (dst = the row of query column j at the
// first channel of the wave's D-quarter), times f
template <int NB>
__device__ __forceinline__ void m2_store_rows(float* dst, int kk, const f4 (&acc)[NB], float f) {
  if constexpr (M2Pool<NB>::WIDE) {
#pragma unroll
    for (int u = 0; u < NB / 4; ++u) {
      float* d = dst + 64 * u + 16 * kk;
      *reinterpret_cast<f4*>(d + 0) = f4{acc[4 * u].x, acc[4 * u + 1].x, acc[4 * u + 2].x, acc[4 * u + 3].x} * f;
      *reinterpret_cast<f4*>(d + 4) = f4{acc[4 * u].y, acc[4 * u + 1].y, acc[4 * u + 2].y, acc[4 * u + 3].y} * f;
      *reinterpret_cast<f4*>(d + 8) = f4{acc[4 * u].z, acc[4 * u + 1].z, acc[4 * u + 2].z, acc[4 * u + 3].z} * f;
      *reinterpret_cast<f4*>(d + 12) = f4{acc[4 * u].w, acc[4 * u + 1].w, acc[4 * u + 2].w, acc[4 * u + 3].w} * f;
    }
  } else {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) *reinterpret_cast<f4*>(dst + 16 * blk + 4 * kk) = acc[blk] * f;
  }
}
// step 3, arithmetic: acc[blk] (16 d x 16 q) += x_tile^T[d][t] * wgt[t][q]
template <int NB>
__device__ __forceinline__ void m2_pool_mfma(const float (&xa)[4][NB], const f4& wgt, f4 (&acc)[NB]) {
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][blk], wgt.x, acc[blk], 0, 0, 0);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1][blk], wgt.y, acc[blk], 0, 0, 0);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[2][blk], wgt.z, acc[blk], 0, 0, 0);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[3][blk], wgt.w, acc[blk], 0, 0, 0);
}
// blocks LO .. LO + CNT - 1 only (the b32 layout): D = 1152 pools its 18 blocks in two halves through ONE set of 36 operand
// registers -- all 72 at once is what made the forward instantiation spill (round 6)
template <int NB, int LO, int CNT>
__device__ __forceinline__ void m2_pool_load_part(const char* tile, const M2Pool<NB>& o, int kq, float (&xa)[4][CNT]) {
  static_assert(!M2Pool<NB>::WIDE, "b32 layout only");
  constexpr int ROWB = 256 * NB;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int b = 0; b < CNT; ++b) {
      const int cb = NB * kq + LO + b;
      const int uni = 4 * s * ROWB + 64 * ((cb & ~3) | ((cb & 3) ^ s));
      xa[s][b] = *reinterpret_cast<const float*>(tile + uni + o.pb[0]);
    }
}
template <int NB, int LO, int CNT>
__device__ __forceinline__ void m2_pool_mfma_part(const float (&xa)[4][CNT], const f4& wgt, f4 (&acc)[NB]) {
#pragma unroll
  for (int b = 0; b < CNT; ++b) acc[LO + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][b], wgt.x, acc[LO + b], 0, 0, 0);
#pragma unroll
  for (int b = 0; b < CNT; ++b) acc[LO + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1][b], wgt.y, acc[LO + b], 0, 0, 0);
#pragma unroll
  for (int b = 0; b < CNT; ++b) acc[LO + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[2][b], wgt.z, acc[LO + b], 0, 0, 0);
#pragma unroll
  for (int b = 0; b < CNT; ++b) acc[LO + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[3][b], wgt.w, acc[LO + b], 0, 0, 0);
}
__device__ __forceinline__ float m2_sel(const f4& v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

// ---------------------------------------------------------------------------------------
// forward.  With EP_MM2_STAGGER = 1 the two halves of the workgroup run STAGGERED: waves 0-3 ("early") finish a tile inside its iteration --
// scores | barrier | softmax | pooling MFMAs -- while waves 4-7 ("late", the SIMD partners of 0-3) carry the pooling
// MFMAs of a tile into the NEXT iteration -- scores | barrier | pooling MFMAs of the previous tile | softmax of this one.
// Between two barriers one wave of every SIMD is then on the matrix pipe while its partner gathers and exponentiates,
// instead of both idling the pipe together.  The late half takes the pooling operands of a tile into registers before the
// barrier that frees the tile's ring slot, so the ring needs no extra slot for it.
// ---------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(M2_NW * 64, 1) void ep_pool_mm2_fwd_kernel(PoolParams p) {
  using C = Mm2Cfg<NG>;
  constexpr int D = C::D, NB = C::NB, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::nslot(false), KD = C::KDMA;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int qb = w >> 2, kq = w & 3;
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory
  const bool n4 = (N & 3) == 0;
  const int tiles_per_img = (N + M2_TT - 1) / M2_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  const int j = lane & 15, kk = lane >> 4;
  const int qj = 16 * qb + j;                // this lane's query (column of the wave's score / pooled blocks)
  const bool late = (EP_MM2_STAGGER == 2 || (EP_MM2_STAGGER == 1 && qb != 0)) && NG <= 6;   // (D >= 896: the carried pooling operands would spill; lockstep there)
  constexpr bool PIPE = EP_MM2_STAGGER == 2 && NG <= 6;
  unsigned long long clk0 = 0, rt0 = 0;
  if constexpr (EP_MM2_CLK != 0) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }

  // B operand of the score MFMAs: query qj over the wave's D-quarter, pre-scaled like the reference (ep.py:39)
  // SLIM (D = 1152, round 6): 72 query + 72 accumulator registers leave no room for the 18 operand offsets and all 72 pooling
  // operands of a tile at two waves per SIMD (the first instantiation spilled 28 registers: 659 us against 568 us as two 16-query
  // chunks).  Here the score offsets are recomputed per tile (3 vector instructions per operand) and the 18 pooled blocks go
  // through ONE set of 36 operand registers in two halves.
  constexpr bool SLIM = NG >= 9;
  float bq[NB][4];
  int aoff[SLIM ? 1 : NB];
#pragma unroll
  for (int g = 0; g < NB; ++g) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (qj < Q) v = *reinterpret_cast<const f4*>(p.cls + (int64_t)qj * D + 16 * NB * kq + 16 * g + 4 * kk);
    v = v * p.scale;
    bq[g][0] = v.x; bq[g][1] = v.y; bq[g][2] = v.z; bq[g][3] = v.w;
    if constexpr (!SLIM) aoff[g] = j * ROWB + (((4 * NB * kq + 4 * g + kk) ^ j) << 4);
  }
  const int arow = j * ROWB, acb = 4 * NB * kq + kk;      // (SLIM: operand g sits at arow + (((acb + 4 g) ^ j) << 4))
  M2Pool<NB> po;
  m2_pool_offsets<NB>(kq, j, kk, po);
  unsigned soff[NG];
  m2_source_offsets<NG>(w, lane, soff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, wg));
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * M2_TT;
      const unsigned limit = (unsigned)((left < M2_TT ? left : M2_TT) * ROWB - 16);
      m2_dma_tile<NG>(psrc, limit, ring + pslot * SLOT, w, soff);
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        const int bn = (wg + pimg * G) < p.B ? (wg + pimg * G) : wg;
        psrc = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, bn));
      } else {
        psrc += SLOT;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 acc[NB];
  float m_j = -INFINITY, mL_j = -INFINITY, lsum = 0.f;     // per lane: running max / partial sum of query qj
  f4 wgt = {0.f, 0.f, 0.f, 0.f};                          // softmax weights in k-slot order (late half: of the pending tile)
  float xa[4][SLIM ? 1 : NB];                             // pooling operands (late half: of the pending tile)
  // online softmax of tile (b, n0) from the summed score block -> wgt; the raw scores go to S (one wave of the block per tile)
  auto softmax = [&](const f4& sc, int b, int n0, int nvalid, int t) {
    if (t == 0) {
      m_j = -INFINITY; mL_j = -INFINITY; lsum = 0.f;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};
    }
    float ue[4] = {sc.x, sc.y, sc.z, sc.w};
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ue[r] = (4 * kk + r) < nvalid ? ue[r] : -INFINITY;
      mx = fmaxf(mx, ue[r]);
    }
    if (__builtin_amdgcn_ballot_w64(mx > m_j + M2_LAZY_MAX_THR) != 0ull) {   // rare
      const float mn = fmaxf(m_j, m2_q4_max(mx));
      const float f = __builtin_amdgcn_exp2f((m_j - mn) * M2_LOG2E);          // m = -inf -> 0
      m_j = mn; mL_j = mn * M2_LOG2E;
      lsum *= f;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) acc[blk] *= f;                        // my column is query qj
    }
    f4 wv;
    wv.x = __builtin_amdgcn_exp2f(fmaf(ue[0], M2_LOG2E, -mL_j));               // invalid tokens: 0
    wv.y = __builtin_amdgcn_exp2f(fmaf(ue[1], M2_LOG2E, -mL_j));
    wv.z = __builtin_amdgcn_exp2f(fmaf(ue[2], M2_LOG2E, -mL_j));
    wv.w = __builtin_amdgcn_exp2f(fmaf(ue[3], M2_LOG2E, -mL_j));
    lsum += (wv.x + wv.y) + (wv.z + wv.w);
    wgt = m2_transpose4(wv);                        // -> the k-slot order of the pooling MFMAs
    if (kq == (t & 3) && qj < Q) {                  // the four waves of the block hold the same scores: they take turns writing them
      float* Srow = p.S + ((int64_t)b * QS + qj) * N + (unsigned)(n0 + 4 * kk);
      if (n4) {
        if (4 * kk < nvalid) *reinterpret_cast<f4*>(Srow) = sc;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((4 * kk + r) < nvalid) Srow[r] = m2_sel(sc, r);
      }
    }
  };

  if (EP_MM2_PRIO != 0 && w >= M2_NW / 2) __builtin_amdgcn_s_setprio(EP_MM2_PRIO);
  // image by image, tile by tile.  One instruction stream for both halves; only the position of the pooling MFMAs differs
  // (late: the pending tile first, early: this tile last); the late half closes an image with the pooling of its last tile
  // before the store (once per image it is not beside the other half's work).  Plain nested loops on purpose: a flat loop with
  // "pending" flags carried around it made the compiler clone the body and spill.
  int cslot = 0, i = 0;
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tp = 0;      // EP_MM2_CLK: cycles per phase of this wave
#define M2_STAMP(k) if constexpr (EP_MM2_CLK != 0) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += tn - tp; tp = tn; }
  if constexpr (EP_MM2_CLK != 0) tp = __builtin_amdgcn_s_memtime();
  for (int img = 0; img < n_img; ++img) {
    const int b = wg + img * G;
    for (int t = 0; t < tiles_per_img; ++t, ++i) {
      const int ahead = pi - 1 - i;
      if (ahead == NSLOT - 2) m2_wait_vmcnt_imm<(NSLOT - 2) * KD>();
      else m2_wait_vmcnt(ahead * KD);
      m2_barrier();                                 // tile i landed everywhere; the slot of tile i-1 is free
      M2_STAMP(0)
      if constexpr (EP_MM2_ABLATE == 5) pi = n_items;  // diagnostic: no ring refills (the arithmetic runs on whatever the first tiles left)
      const int n0 = t * M2_TT;
      const int nvalid = (N - n0) < M2_TT ? (N - n0) : M2_TT;
      const char* tile = ring + cslot * SLOT;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if constexpr (EP_MM2_ABLATE == 1) { produce(); m2_barrier(); continue; }
      if constexpr (SLIM) {
        int cbv = acb;
        asm volatile("" : "+v"(cbv));               // (opaque per tile: keeps the 18 offsets out of registers)
        m2_scores<NB>(tile, [&](int g) { return arow + (((cbv + 4 * g) ^ j) << 4); }, bq, spart, w, lane, produce);
        m2_barrier();                               // all partial score blocks are in the scratch
        const f4 sc = m2_gather(spart, qb, lane);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int HB = NB / 2;
        float xh[4][HB];
        m2_pool_load_part<NB, 0, HB>(tile, po, kq, xh);          // in flight under the softmax arithmetic
        softmax(sc, b, n0, nvalid, t);
        m2_pool_mfma_part<NB, 0, HB>(xh, wgt, acc);
        m2_pool_load_part<NB, HB, NB - HB>(tile, po, kq, xh);    // (the slot stays valid until the next iteration's barrier)
        m2_pool_mfma_part<NB, HB, NB - HB>(xh, wgt, acc);
      }
      if constexpr (!SLIM) {
      m2_scores<NB>(tile, [&](int g) { return aoff[g]; }, bq, spart, w, lane, produce);
      M2_STAMP(1)
      if constexpr (PIPE) { if (t > 0) m2_pool_mfma<NB>(xa, wgt, acc); }       // the pending tile: one matrix phase per iteration
      m2_barrier();                                 // all partial score blocks are in the scratch
      M2_STAMP(2)
      if constexpr (!PIPE) { if (late && t > 0) m2_pool_mfma<NB>(xa, wgt, acc); }
      __builtin_amdgcn_sched_barrier(0);
      M2_STAMP(3)
      if constexpr (EP_MM2_ABLATE == 4 || EP_MM2_ABLATE == 5) {
        wgt = f4{0.001f * lane, 0.002f, 0.003f, 0.004f};
        m2_pool_load<NB>(tile, po, kq, xa);
      } else {
        const f4 sc = m2_gather(spart, qb, lane);
        __builtin_amdgcn_sched_barrier(0);
        m2_pool_load<NB>(tile, po, kq, xa);      // in flight under the softmax arithmetic; complete before the next barrier
        softmax(sc, b, n0, nvalid, t);
      }
      M2_STAMP(4)
      if (!late) m2_pool_mfma<NB>(xa, wgt, acc);
      M2_STAMP(5)
      }
    }
    if constexpr (!SLIM) { if (late) m2_pool_mfma<NB>(xa, wgt, acc); }
    if constexpr (EP_MM2_ABLATE != 1) {             // image b is complete in acc / m_j / lsum: normalise and store
      const float l = m2_q4_sum(lsum);
      const float inv = 1.0f / l;
      if (qj < Q) {
        m2_store_rows<NB>(p.P + ((int64_t)b * QS + qj) * D + 16 * NB * kq, kk, acc, inv);
        if (kq == 0 && kk == 0) {
          const f4 rec = {m_j, l, 0.f, 0.f};
          *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + qj) * 4) = rec;
        }
      }
    }
  }
  if constexpr (EP_MM2_CLK != 0) {
    if (p.dbg && threadIdx.x == 0) {
      p.dbg[(int64_t)wg * 2] = __builtin_amdgcn_s_memtime() - clk0;
      p.dbg[(int64_t)wg * 2 + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    if (p.dbg && wg == 0 && lane == 0)
      for (int k = 0; k < 6; ++k) p.dbg[1024 + w * 8 + k] = ph[k];
  }
}

// ---------------------------------------------------------------------------------------
// backward.  Ring items per image: two header tiles with the rows 0-15 / 16-31 of dP[b] (B operands of the dA MFMAs of query
// block 0 / 1), then the token tiles.  Every item also carries one 4-byte-per-lane DMA per wave (keeps the counted waits
// uniform): header item h -> the ML rows of block h (16 queries x 4 floats), token tile -> S[b, q, n0:n0+16] (waves 0-3:
// queries 0-15, waves 4-7: queries 16-31; wave kq copies queries 4 kq .. 4 kq + 3 of its block).
// Staggered like the forward: the late half runs the pooling MFMAs of a token tile in the next tile's iteration.
// ---------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(M2_NW * 64, 1) void ep_pool_mm2_bwd_kernel(PoolParams p) {
  using C = Mm2Cfg<NG>;
  constexpr int D = C::D, NB = C::NB, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::nslot(true), KD = C::KDMA + 1, SMALLB = C::SMALL;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  char* small_base = spart + C::SPART;                  // [NSLOT][SMALLB]
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int qb = w >> 2, kq = w & 3;
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;
  const int tiles_per_img = (N + M2_TT - 1) / M2_TT;
  const int items_per_img = 2 + tiles_per_img;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * items_per_img;
  const int j = lane & 15, kk = lane >> 4;
  const int qj = 16 * qb + j;
  const bool live = qj < Q;
  const bool late = (EP_MM2_STAGGER == 2 || (EP_MM2_STAGGER == 1 && qb != 0)) && NG <= 6;   // (D >= 896: the carried pooling operands would spill; lockstep there)
  constexpr bool PIPE = EP_MM2_STAGGER == 2 && NG <= 6;

  f4 gacc[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) gacc[blk] = f4{0.f, 0.f, 0.f, 0.f};

  if (n_items > 0) {
    int aoff[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) aoff[g] = j * ROWB + (((4 * NB * kq + 4 * g + kk) ^ j) << 4);
    M2Pool<NB> po;
    m2_pool_offsets<NB>(kq, j, kk, po);
    unsigned soff[NG];
    m2_source_offsets<NG>(w, lane, soff);
    // small DMA of a token item: wave (qb, kq) copies elements E = 64*kq + lane of the 16x16 block S[b, 16 qb + (E>>4), n0 + (E&15)]
    const int se = 64 * kq + lane;
    const int sq = (16 * qb + (se >> 4)) < Q ? (16 * qb + (se >> 4)) : Q - 1;

    int pi = 0, pimg = 0, pidx = 0, pslot = 0;
    auto produce = [&]() {
      if (pi < n_items) {
        const int b = wg + pimg * G;
        char* slot = ring + pslot * SLOT;
        char* small = small_base + pslot * SMALLB;
        if (pidx < 2) {
          const int left = Q - 16 * pidx;                                          // >= 1: this kernel runs 17 .. 32 queries
          const int rows = left < M2_TT ? left : M2_TT;
          const char* src = reinterpret_cast<const char*>(p.dP + ((int64_t)b * QS + 16 * pidx) * D);
          m2_dma_tile<NG>(src, (unsigned)(rows * ROWB - 16), slot, w, soff);
          const int hq = (16 * pidx + (lane >> 2)) < Q ? (16 * pidx + (lane >> 2)) : Q - 1;      // ML[b, 16 h + lane / 4, lane % 4]
          const float* ms = p.ML + ((int64_t)b * QS + hq) * 4 + (lane & 3);        // (all waves copy the same 256 bytes)
          __builtin_amdgcn_global_load_lds((m2_gptr_t)ms, (m2_lds_ptr_t)(small + 2048), 4, 0, 0);
        } else {
          const int n0 = (pidx - 2) * M2_TT;
          const int rows = (N - n0) < M2_TT ? (N - n0) : M2_TT;
          const char* src = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, b) + (int64_t)n0 * D);
          m2_dma_tile<NG>(src, (unsigned)(rows * ROWB - 16), slot, w, soff);
          int nn = n0 + (se & 15); nn = nn < N ? nn : N - 1;
          const float* ss = p.S + ((int64_t)b * QS + sq) * N + nn;
          __builtin_amdgcn_global_load_lds((m2_gptr_t)ss, (m2_lds_ptr_t)(small + 1024 * qb + 256 * kq), 4, 0, 0);
        }
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++pidx == items_per_img) { pidx = 0; ++pimg; }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    float bq[NB][4];
    float mL_j = 0.f, il_j = 0.f, dl_j = 0.f;
    f4 wgt = {0.f, 0.f, 0.f, 0.f};
    float xa[4][NB];                                    // (late half: operands of the pending tile)
    if (EP_MM2_PRIO != 0 && w >= M2_NW / 2) __builtin_amdgcn_s_setprio(EP_MM2_PRIO);
    int cslot = 0, i = 0;
    auto next_item = [&](const char*& tile, const char*& small) {
      const int ahead = pi - 1 - i;
      if (ahead == NSLOT - 2) m2_wait_vmcnt_imm<(NSLOT - 2) * KD>();
      else m2_wait_vmcnt(ahead * KD);
      m2_barrier();
      tile = ring + cslot * SLOT;
      small = small_base + cslot * SMALLB;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      ++i;
    };
    for (int img = 0; img < n_img; ++img) {
      const char* tile; const char* small;
      // header items: item h holds rows 16 h .. 16 h + 15 of dP[b] and the ML rows of block h; a wave takes its own block's
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        next_item(tile, small);
        produce();
        if (h == qb) {
#pragma unroll
          for (int g = 0; g < NB; ++g) {
            f4 v = *reinterpret_cast<const f4*>(tile + aoff[g]);
            if (!live) v = f4{0.f, 0.f, 0.f, 0.f};
            bq[g][0] = v.x; bq[g][1] = v.y; bq[g][2] = v.z; bq[g][3] = v.w;
          }
          const f4 rec = *reinterpret_cast<const f4*>(small + 2048 + 16 * j);        // (rows beyond Q hold the last query's, from the clamped copy)
          mL_j = rec.x * M2_LOG2E; il_j = 1.0f / rec.y; dl_j = rec.z;
        }
      }
      for (int t = 0; t < tiles_per_img; ++t) {
        next_item(tile, small);
        const int n0 = t * M2_TT;
        const int nvalid = (N - n0) < M2_TT ? (N - n0) : M2_TT;
        m2_scores<NB>(tile, [&](int g) { return aoff[g]; }, bq, spart, w, lane, produce);      // dA partial blocks
        if constexpr (PIPE) { if (t > 0) m2_pool_mfma<NB>(xa, wgt, gacc); }
        m2_barrier();
        if constexpr (!PIPE) { if (late && t > 0) m2_pool_mfma<NB>(xa, wgt, gacc); }
        __builtin_amdgcn_sched_barrier(0);
        const f4 u = m2_gather(spart, qb, lane);
        __builtin_amdgcn_sched_barrier(0);
        m2_pool_load<NB>(tile, po, kq, xa);
        {
          // dS of the tile = A (dA - delta) from the summed dA block and the saved scores S[b, qj, n0 + 4 kk + {0..3}]
          const f4 sv = *reinterpret_cast<const f4*>(small + 1024 * qb + 4 * (16 * j + 4 * kk));
          const float a0 = __builtin_amdgcn_exp2f(fmaf(sv.x, M2_LOG2E, -mL_j)) * il_j;
          const float a1 = __builtin_amdgcn_exp2f(fmaf(sv.y, M2_LOG2E, -mL_j)) * il_j;
          const float a2 = __builtin_amdgcn_exp2f(fmaf(sv.z, M2_LOG2E, -mL_j)) * il_j;
          const float a3 = __builtin_amdgcn_exp2f(fmaf(sv.w, M2_LOG2E, -mL_j)) * il_j;
          f4 wv;
          wv.x = ((4 * kk + 0) < nvalid && live) ? a0 * (u.x - dl_j) : 0.f;
          wv.y = ((4 * kk + 1) < nvalid && live) ? a1 * (u.y - dl_j) : 0.f;
          wv.z = ((4 * kk + 2) < nvalid && live) ? a2 * (u.z - dl_j) : 0.f;
          wv.w = ((4 * kk + 3) < nvalid && live) ? a3 * (u.w - dl_j) : 0.f;
          wgt = m2_transpose4(wv);                  // -> the k-slot order of the pooling MFMAs
        }
        if (!late) m2_pool_mfma<NB>(xa, wgt, gacc);
      }
      if (late) m2_pool_mfma<NB>(xa, wgt, gacc);                      // the image's last tile
    }
  }
  if (live) {
    m2_store_rows<NB>(p.Gpart + ((int64_t)wg * Q + qj) * D + 16 * NB * kq, kk, gacc, 1.0f);
  }
}

// ---------------------------------------------------------------------------------------
template <int NG>
static int mm2_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  using C = Mm2Cfg<NG>;
  static_assert(C::nslot(false) >= 2 && C::nslot(true) >= 2, "32-query matrix-core pooling: the ring does not fit");
  const size_t lds = C::lds_bytes(bwd);
  if (bwd) {
    auto kb = ep_pool_mm2_bwd_kernel<NG>;
    hipError_t e = hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(kb, dim3(grid), dim3(M2_NW * 64), lds, st, p);
    EP_LAUNCH_CHECK("ep_pool_mm2_bwd_kernel");
    return 0;
  }
  auto kf = ep_pool_mm2_fwd_kernel<NG>;
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
  if constexpr (EP_MM2_CLK != 0) {
    static unsigned long long* dbg = nullptr;
    static int calls = 0;
    if (!dbg) (void)hipMalloc(&dbg, 4096 * 2 * sizeof(unsigned long long));
    PoolParams q = p;
    q.dbg = dbg;
    hipLaunchKernelGGL(kf, dim3(grid), dim3(M2_NW * 64), lds, st, q);
    if (++calls == 20) {
      unsigned long long host[2 * 8], phs[64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
      (void)hipMemcpy(phs, dbg + 1024, sizeof(phs), hipMemcpyDeviceToHost);
      for (int wv = 0; wv < 8; ++wv)
        fprintf(stderr, "[mm2 phases] wave %d: wait+barrier1 %llu | scores %llu | barrier2 %llu | late pool %llu | gather+softmax %llu | early pool %llu\n", wv,
                phs[wv * 8], phs[wv * 8 + 1], phs[wv * 8 + 2], phs[wv * 8 + 3], phs[wv * 8 + 4], phs[wv * 8 + 5]);
      for (int k = 0; k < 4; ++k)
        fprintf(stderr, "[mm2 clk] wg %d: %llu shader cycles in %.1f us -> %.0f MHz\n", k, host[2 * k], host[2 * k + 1] / 100.0,
                host[2 * k + 1] ? 100.0 * (double)host[2 * k] / (double)host[2 * k + 1] : 0.0);
    }
    return 0;
  }
  hipLaunchKernelGGL(kf, dim3(grid), dim3(M2_NW * 64), lds, st, p);
  EP_LAUNCH_CHECK("ep_pool_mm2_fwd_kernel");
  return 0;
}

// fp32 tokens, shared query rows, 17 .. 32 queries, D = 128 k up to 1024 (two ring slots + the per-slot score / statistics
// pieces must fit the 160 KiB, the register budget is 256 per wave at two waves per SIMD)
bool mm2_supported(int D, int Q, int64_t cls_bstride, bool bwd) {
  if (cls_bstride != 0 || Q <= 16 || Q > 32 || D % 128 != 0 || D < 256) return false;
  // D = 1152 (SigLIP2 SO400M, round 6): the BACKWARD only -- 244 registers, 409 us in the step at 1024 x 256 against 586 us as two
  // 16-query chunks; the forward instantiation spills 28 registers (256 + scratch) and is SLOWER than the chunks (659 against 568 us)
  // (... round 6, later: the SLIM forward -- offsets recomputed, pooling operands in two halves: 233 registers -- fits: 256 x 1152 at 32
  // queries, same box, 1.262 -> 1.117 ms per step; EP_POOL_MM2_FWD9=0: chunks)
  static int fwd9 = -1;
  if (fwd9 < 0) { const char* e = getenv("EP_POOL_MM2_FWD9"); fwd9 = e ? atoi(e) : 1; }
  return D <= 1024 || (D == 1152 && (bwd || fwd9));
}

int mm2_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  switch (p.D / 128) {
    case 2: return mm2_launch_one<2>(bwd, p, grid, st);
    case 3: return mm2_launch_one<3>(bwd, p, grid, st);
    case 4: return mm2_launch_one<4>(bwd, p, grid, st);
    case 5: return mm2_launch_one<5>(bwd, p, grid, st);
    case 6: return mm2_launch_one<6>(bwd, p, grid, st);
    case 7: return mm2_launch_one<7>(bwd, p, grid, st);
    case 8: return mm2_launch_one<8>(bwd, p, grid, st);
    case 9: return mm2_launch_one<9>(bwd, p, grid, st);      // round 6: SigLIP2 SO400M's 1152 (ring of two 72 KiB tiles, as D = 1024)
  }
  set_error("no 32-query matrix-core pooling kernel for D=%d", p.D);
  return EP_E_UNSUPPORTED;
}

}  // namespace ep
