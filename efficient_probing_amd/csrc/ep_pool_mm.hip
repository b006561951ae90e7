// EP attentive pooling, all-matrix-core variant (gfx950 / CDNA4): BOTH contractions of a token tile
// run on v_mfma_f32_16x16x4_f32 (exact fp32, bit-identical to an fmaf chain):
//
//   scores   S[t][q]   = sum_d x[t][d] * (cls[q][d]*scale)        A = x tile,  B = queries
//   pooling  P^T[d][q] = sum_t x[t][d] * softmax-weight[q][t]     A = x tile^T, B = weights
//
// (reference poolings/ep.py:35-44 forward; the backward is the same pair with dP in place of the
// queries and A*(dA-delta) in place of the softmax weights.)
//
// One 8-wave workgroup per CU streams whole images through a ring of 16-token tiles filled by
// LDS-DMA; wave w owns the D-slice [D/8*w, D/8*(w+1)) for BOTH contractions and for ALL queries:
//   1. score partials over its slice (D/32 MFMAs), summed across the 8 waves through a padded LDS
//      scratch; every wave then holds the full 16x16 score block in the MFMA B layout
//      (lane = (token mod 4, query), 4 token groups in 4 registers) -- exactly the operand layout of
//      step 3, so the softmax weights never move between lanes;
//   2. lazy-max online softmax on 4 values per lane (per-lane state of query lane&15);
//   3. pooling of its slice for all 16 query columns (D/32 MFMAs), accumulators in the MFMA D layout
//      (column = query, rows = 16 consecutive d) -- a max-rescale is a per-lane multiply.
// The tile is stored XOR-swizzled by row (free, on the DMA source address): the b128 row-strided
// reads of step 1 are conflict-free, the b32 reads of step 3 at most 2-way.
// Per tile and wave: D/16 MFMAs, ~D/32 + 4 LDS reads, ~60 vector instructions.
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"

namespace ep {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

#ifndef EP_MM_ABLATE
#define EP_MM_ABLATE 0                // diagnostic builds of the forward (tools/build_variant.sh): 1 ring + barriers only,
#endif                                // 2 + score MFMAs, 3 + gather / softmax (no pooling MFMAs); results are then wrong
constexpr int MM_TT = 16;             // tokens per tile
constexpr int MM_NW = 8;              // waves per workgroup (one workgroup per CU)
constexpr float MM_LOG2E = 1.4426950408889634f;
constexpr float MM_LAZY_MAX_THR = 12.0f;

template <int NG>
struct MmCfg {
  static constexpr int D = 128 * NG;
  static constexpr int ROWB = 4 * D;
  static constexpr int SLOT = MM_TT * ROWB;
  static constexpr int KDMA = NG;                    // 1 KiB DMA pieces per wave per tile
  static constexpr int SPART = MM_NW * 272 * 4;      // partial score blocks (padded: conflict-free gather)
  static constexpr int SMALL = 2048;                 // per slot (backward): S tile (16 q x 16 t floats) + ML rows (64 x 16 B)
  static constexpr int TS = 256;                     // per slot (LayerNorm-of-tokens mode): {mean, rstd} of the 16 tokens
  static constexpr int LDS_TOTAL = 160 * 1024;
  static constexpr int nslot(bool bwd, bool ln) {
    int ns = (LDS_TOTAL - SPART) / (SLOT + (bwd ? SMALL : 0) + (ln ? TS : 0));
    return ns > 4 ? 4 : ns;
  }
  static constexpr bool VALID = nslot(false, true) >= 2 && nslot(true, true) >= 2;
};

__device__ __forceinline__ void mm_wait_vmcnt(int n) {
#define EP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    EP_W(0) EP_W(1) EP_W(2) EP_W(3) EP_W(4) EP_W(5) EP_W(6) EP_W(7) EP_W(8) EP_W(9)
    EP_W(10) EP_W(11) EP_W(12) EP_W(13) EP_W(14) EP_W(15) EP_W(16) EP_W(17) EP_W(18) EP_W(19)
    EP_W(20) EP_W(21) EP_W(22) EP_W(23) EP_W(24) EP_W(25) EP_W(26) EP_W(27) EP_W(28) EP_W(29)
    EP_W(30) EP_W(31) EP_W(32) EP_W(33)
    default: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
  }
#undef EP_W
}
template <int N>
__device__ __forceinline__ void mm_wait_vmcnt_imm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate out of range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void mm_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// combine a per-lane value over the 4 lanes that share a query (lane, lane^16, lane^32, lane^48)
__device__ __forceinline__ float q4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float q4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int NG>
__device__ __forceinline__ void mm_source_offsets(int w, int lane, unsigned (&soff)[NG]) {
  constexpr int NCHUNK = 32 * NG;
#pragma unroll
  for (int jj = 0; jj < NG; ++jj) {
    const int pos = (w + MM_NW * jj) * 64 + lane;
    const int t = pos / NCHUNK, c = pos - t * NCHUNK;
    soff[jj] = (unsigned)(t * (512 * NG) + ((c ^ (t & 15)) << 4));
  }
}
template <int NG>
__device__ __forceinline__ void mm_dma_tile(const char* src, unsigned limit, char* slot, int w,
                                            const unsigned (&soff)[NG]) {
#pragma unroll
  for (int jj = 0; jj < NG; ++jj) {
    const unsigned off = soff[jj] < limit ? soff[jj] : limit;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lds_ptr_t)(slot + (w + MM_NW * jj) * 1024), 16, 0,
                                     EP_DMA_AUX);
  }
}

__device__ __forceinline__ void continue_tile(int& ctile, int& cimg, int tiles_per_img) {   // (ablation builds only)
  if (ctile == tiles_per_img - 1) { ctile = 0; ++cimg; } else ++ctile;
}
// step 1: 16 tokens x 16 queries over this wave's D-slice -> LDS scratch; `mid` (the ring refill) is
// issued in the shadow of the first MFMAs.  Two accumulators halve the dependent-chain latency.
template <int NG, typename F>
__device__ __forceinline__ void mm_scores(const char* tile, const int (&aoff)[NG], const float (&bq)[NG][4],
                                          char* spart, int w, int lane, F&& mid) {
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f4 xa[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) xa[g] = *reinterpret_cast<const f4*>(tile + aoff[g]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].x, bq[g][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].y, bq[g][1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].z, bq[g][2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].w, bq[g][3], acc1, 0, 0, 0);
    if (g == 0) mid();
  }
  *reinterpret_cast<f4*>(spart + (w * 272 + lane * 4 + (lane >> 4) * 4) * 4) = acc0 + acc1;
}
// full score of (query j = lane & 15, token 4s + kk): D layout col = query, row = token ->
// lane 16*s + j, register kk of every wave's block  (conflict-free: 68 s + 4 j + kk)
__device__ __forceinline__ void mm_gather(const char* spart, int j, int kk, float (&s)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* base = reinterpret_cast<const float*>(spart) + (16 * g + j) * 4 + g * 4 + kk;
    float v = 0.f;
#pragma unroll
    for (int ws = 0; ws < MM_NW; ++ws) v += base[ws * 272];
    s[g] = v;
  }
}
// step 3: acc[blk] (16 d x 16 q) += x_tile^T[d][t] * wgt[t][q]; A operand lane (i = d, kk = token mod 4).
// Element x[t = 4s + kk][16*NG*w + 16*blk + i] lives in chunk c = 4*(NG*w + blk) + (i >> 2) of row t, stored at
// chunk position c ^ (t & 15) = 4*(cb with its low 2 bits ^ s) + ((i >> 2) ^ kk), cb = NG*w + blk:
//   byte offset = [kk*ROWB + 16*((i>>2)^kk) + 4*(i&3)]  (per lane: `plane`)  +  4*s*ROWB + 64*swz(cb, s)  (wave-uniform)
template <int NG>
__device__ __forceinline__ void mm_pool(const char* tile, int plane, int w, const float (&wgt)[4], f4 (&acc)[NG]) {
  constexpr int ROWB = 512 * NG;
  float xa[4][NG];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int blk = 0; blk < NG; ++blk) {
      const int cb = NG * w + blk;
      const int uni = 4 * s * ROWB + 64 * ((cb & ~3) | ((cb & 3) ^ s));
      xa[s][blk] = *reinterpret_cast<const float*>(tile + uni + plane);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int blk = 0; blk < NG; ++blk)
      acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s][blk], wgt[s], acc[blk], 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// LN: LayerNorm-of-tokens mode (PoolParams.tokstat): the pass pools xhat = (x - mean) rstd without materialising it --
// scores rstd (q.x - mean sum(q)), pooling weights a rstd, one scalar per query (sum_n a rstd mean) taken off at the end;
// the {mean, rstd} pairs of a tile ride in the ring as one more 4-byte-per-lane DMA piece per wave.
template <int NG, bool LN>
__global__ __launch_bounds__(MM_NW * 64, 2) void ep_pool_mm_fwd_kernel(PoolParams p) {
  using C = MmCfg<NG>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::nslot(false, LN), KDMA = C::KDMA;
  constexpr int KD = KDMA + (LN ? 1 : 0);
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  char* tsbase = spart + C::SPART;                  // LN: [NSLOT][256 B]
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const int tiles_per_img = (N + MM_TT - 1) / MM_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  const int j = lane & 15, kk = lane >> 4;

  // B operand of the score MFMAs: queries pre-scaled like the reference (ep.py:39)
  float bq[NG][4];
  int aoff[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < Q) v = *reinterpret_cast<const f4*>(p.cls + (int64_t)j * D + 16 * NG * w + 16 * g + 4 * kk);
    v = v * p.scale;
    bq[g][0] = v.x; bq[g][1] = v.y; bq[g][2] = v.z; bq[g][3] = v.w;
    aoff[g] = j * ROWB + (((4 * NG * w + 4 * g + kk) ^ j) << 4);
  }
  // per-lane part of the pooling A-operand address (see mm_pool)
  const int plane = kk * ROWB + 16 * ((j >> 2) ^ kk) + 4 * (j & 3);
  unsigned soff[NG];
  mm_source_offsets<NG>(w, lane, soff);
  float wsum_j = 0.f;                                // LN: sum over D of the scaled query j
  if (LN) {
    if (j < Q)
      for (int c = kk; c < D / 4; c += 4) {
        const f4 v = *reinterpret_cast<const f4*>(p.cls + (int64_t)j * D + 4 * c) * p.scale;
        wsum_j += (v.x + v.y) + (v.z + v.w);
      }
    wsum_j = q4_sum(wsum_j);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, wg));
  const float* pts = LN ? p.tokstat + (int64_t)(p.index ? p.index[wg] : wg) * N * 2 : nullptr;
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * MM_TT;
      const unsigned limit = (unsigned)((left < MM_TT ? left : MM_TT) * ROWB - 16);
      mm_dma_tile<NG>(psrc, limit, ring + pslot * SLOT, w, soff);
      if (LN) {                                      // every wave copies the same 128 bytes: keeps the counted waits uniform
        int e = ptile * MM_TT * 2 + (lane & 31); e = e < 2 * N ? e : 2 * N - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(pts + e), (lds_ptr_t)(tsbase + pslot * C::TS), 4, 0, 0);
      }
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        const int bn = (wg + pimg * G) < p.B ? (wg + pimg * G) : wg;
        psrc = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, bn));
        if (LN) pts = p.tokstat + (int64_t)(p.index ? p.index[bn] : bn) * N * 2;
      } else {
        psrc += SLOT;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 acc[NG];
  float m_j = -INFINITY, mL_j = -INFINITY, lsum = 0.f;     // per lane: running max / partial sum of query j
  float c2 = 0.f;                                          // LN: partial of sum_n a rstd mean
  int cimg = 0, ctile = 0, cslot = 0;
  for (int i = 0; i < n_items; ++i) {
    const int ahead = pi - 1 - i;
    if (ahead == NSLOT - 2) mm_wait_vmcnt_imm<(NSLOT - 2) * KD>();
    else mm_wait_vmcnt(ahead * KD);
    mm_barrier();                                   // tile i landed everywhere; slot of tile i-1 is free
    const int b = wg + cimg * G;
    const int n0 = ctile * MM_TT;
    const int nvalid = (N - n0) < MM_TT ? (N - n0) : MM_TT;
    const char* tile = ring + cslot * SLOT;
    const float* tstat = reinterpret_cast<const float*>(tsbase + cslot * C::TS);
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    if (ctile == 0) {
      m_j = -INFINITY; mL_j = -INFINITY; lsum = 0.f; c2 = 0.f;
#pragma unroll
      for (int blk = 0; blk < NG; ++blk) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (EP_MM_ABLATE == 1) { produce(); mm_barrier(); continue_tile(ctile, cimg, tiles_per_img); continue; }
    mm_scores<NG>(tile, aoff, bq, spart, w, lane, produce);
    mm_barrier();                                   // all partial score blocks are in the scratch
    if constexpr (EP_MM_ABLATE == 2) { continue_tile(ctile, cimg, tiles_per_img); continue; }
    float sc[4], ue[4];
    mm_gather(spart, j, kk, sc);
    float tm[4], tr[4];
    if (LN) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        tm[s] = tstat[2 * (4 * s + kk)]; tr[s] = tstat[2 * (4 * s + kk) + 1];
        sc[s] = tr[s] * (sc[s] - tm[s] * wsum_j);                             // q . xhat
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ue[s] = (4 * s + kk) < nvalid ? sc[s] : -INFINITY;
      mx = fmaxf(mx, ue[s]);
    }
    if (__builtin_amdgcn_ballot_w64(mx > m_j + MM_LAZY_MAX_THR) != 0ull) {    // rare
      const float mn = fmaxf(m_j, q4_max(mx));
      const float f = __builtin_amdgcn_exp2f((m_j - mn) * MM_LOG2E);           // m = -inf -> 0
      m_j = mn; mL_j = mn * MM_LOG2E;
      lsum *= f; c2 *= f;
#pragma unroll
      for (int blk = 0; blk < NG; ++blk) acc[blk] *= f;                       // my column is query j
    }
    float wgt[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wgt[s] = __builtin_amdgcn_exp2f(fmaf(ue[s], MM_LOG2E, -mL_j));          // invalid tokens: 0
      lsum += wgt[s];
      if (LN) { wgt[s] *= tr[s]; c2 = fmaf(wgt[s], tm[s], c2); }              // pooling weights a rstd
    }
    if (w == 0 && j < Q) {                          // every wave holds the same scores: one writes them
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if ((4 * s + kk) < nvalid) p.S[((int64_t)b * QS + j) * N + (unsigned)(n0 + 4 * s + kk)] = sc[s];
    }
    if constexpr (EP_MM_ABLATE != 3) mm_pool<NG>(tile, plane, w, wgt, acc);
    if (ctile == tiles_per_img - 1) {
      const float l = q4_sum(lsum);
      const float inv = 1.0f / l;
      const float shift = LN ? q4_sum(c2) * inv : 0.f;                        // sum_n A rstd_n mean_n
      if (j < Q) {
        float* Pq = p.P + ((int64_t)b * QS + j) * D + 16 * NG * w + 4 * kk;      // rows i = 4*kk + r of each block
#pragma unroll
        for (int blk = 0; blk < NG; ++blk) *reinterpret_cast<f4*>(Pq + 16 * blk) = acc[blk] * inv - shift;
        if (w == 0 && kk == 0) {
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
// backward.  Ring items per image: one header tile with the Q rows of dP[b], then the token
// tiles.  Every item also carries one 16-byte-per-lane DMA per wave (all waves copy the same 1 KiB:
// benign duplicates that keep the counted waits uniform): header -> ML[b, q, 0:4] rows, token
// tile -> S[b, q, n0:n0+16] (lane = query*4 + quarter).
// ---------------------------------------------------------------------------------------
template <int NG, bool LN>
__global__ __launch_bounds__(MM_NW * 64, 2) void ep_pool_mm_bwd_kernel(PoolParams p) {
  using C = MmCfg<NG>;
  constexpr int D = C::D, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::nslot(true, LN), KDMA = C::KDMA;
  constexpr int KD = KDMA + 1;
  constexpr int SMALLB = C::SMALL + (LN ? C::TS : 0);   // per slot: S tile | ML rows | LN: {mean, rstd} of the tile
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  char* small_base = spart + C::SPART;                  // [NSLOT][SMALLB]
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const int tiles_per_img = (N + MM_TT - 1) / MM_TT;
  const int items_per_img = 1 + tiles_per_img;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * items_per_img;
  const int j = lane & 15, kk = lane >> 4;

  f4 gacc[NG];
  float c3 = 0.f;                                       // LN: partial of sum_b sum_n dS rstd mean
#pragma unroll
  for (int blk = 0; blk < NG; ++blk) gacc[blk] = f4{0.f, 0.f, 0.f, 0.f};

  if (n_items > 0) {
    int aoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) aoff[g] = j * ROWB + (((4 * NG * w + 4 * g + kk) ^ j) << 4);
    const int plane = kk * ROWB + 16 * ((j >> 2) ^ kk) + 4 * (j & 3);
    unsigned soff[NG];
    mm_source_offsets<NG>(w, lane, soff);
    // small DMA of a token item: wave w copies elements E = 64*(w&3) + lane of the 16x16 block S[b, E>>4, n0 + (E&15)]
    const int se = 64 * (w & 3) + lane;
    const int sq = (se >> 4) < Q ? (se >> 4) : Q - 1;

    int pi = 0, pimg = 0, pidx = 0, pslot = 0;
    auto produce = [&]() {
      if (pi < n_items) {
        const int b = wg + pimg * G;
        char* slot = ring + pslot * SLOT;
        char* small = small_base + pslot * SMALLB;
        if (pidx == 0) {
          const char* src = reinterpret_cast<const char*>(p.dP + (int64_t)b * QS * D);
          const int rows = Q < MM_TT ? Q : MM_TT;
          mm_dma_tile<NG>(src, (unsigned)(rows * ROWB - 16), slot, w, soff);
          int hq = lane < Q ? lane : Q - 1;                                       // ML[b, lane, 0:4]
          const float* ms = p.ML + ((int64_t)b * QS + hq) * 4;
          __builtin_amdgcn_global_load_lds((gptr_t)ms, (lds_ptr_t)(small + 1024), 16, 0, 0);   // lanes 0-15 matter
        } else {
          const int n0 = (pidx - 1) * MM_TT;
          const int rows = (N - n0) < MM_TT ? (N - n0) : MM_TT;
          const char* src = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, b) + (int64_t)n0 * D);
          mm_dma_tile<NG>(src, (unsigned)(rows * ROWB - 16), slot, w, soff);
          int nn = n0 + (se & 15); nn = nn < N ? nn : N - 1;
          const float* ss = p.S + ((int64_t)b * QS + sq) * N + nn;
          char* sdst = small + 256 * (w & 3);
          if (LN && w >= 4) {                         // waves 4-7 (duplicates of 0-3 otherwise) fetch the tile's {mean, rstd}
            int e = n0 * 2 + (lane & 31); e = e < 2 * N ? e : 2 * N - 1;
            ss = p.tokstat + (int64_t)(p.index ? p.index[b] : b) * N * 2 + e;
            sdst = small + C::SMALL;
          }
          __builtin_amdgcn_global_load_lds((gptr_t)ss, (lds_ptr_t)sdst, 4, 0, 0);
        }
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++pidx == items_per_img) { pidx = 0; ++pimg; }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    float bq[NG][4];
    float mL_j = 0.f, il_j = 0.f, dl_j = 0.f;
    float gsum_j = 0.f;                                 // LN: sum over D of dP[b, j]
    int cidx = 0, cslot = 0;
    for (int i = 0; i < n_items; ++i) {
      const int ahead = pi - 1 - i;
      if (ahead == NSLOT - 2) mm_wait_vmcnt_imm<(NSLOT - 2) * KD>();
      else mm_wait_vmcnt(ahead * KD);
      mm_barrier();
      const char* tile = ring + cslot * SLOT;
      const char* small = small_base + cslot * SMALLB;
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if (cidx == 0) {
        produce();
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          f4 v = *reinterpret_cast<const f4*>(tile + aoff[g]);
          if (j >= Q) v = f4{0.f, 0.f, 0.f, 0.f};
          bq[g][0] = v.x; bq[g][1] = v.y; bq[g][2] = v.z; bq[g][3] = v.w;
        }
        const f4 rec = *reinterpret_cast<const f4*>(small + 1024 + 16 * (j < Q ? j : Q - 1));
        mL_j = rec.x * MM_LOG2E; il_j = 1.0f / rec.y; dl_j = rec.z;
        if (LN) {
          // sum of the whole dP row: slice sums of the 8 waves through the score scratch (free between two token items:
          // the barrier at the top of this item is behind every wave's gather of the previous one)
          float gs = 0.f;
#pragma unroll
          for (int g = 0; g < NG; ++g) gs += (bq[g][0] + bq[g][1]) + (bq[g][2] + bq[g][3]);
          gs = q4_sum(gs);
          float* sp = reinterpret_cast<float*>(spart);
          if (kk == 0) sp[w * 16 + j] = gs;
          mm_barrier();
          gsum_j = 0.f;
#pragma unroll
          for (int ws = 0; ws < MM_NW; ++ws) gsum_j += sp[ws * 16 + j];
        }
      } else {
        const int n0 = (cidx - 1) * MM_TT;
        const int nvalid = (N - n0) < MM_TT ? (N - n0) : MM_TT;
        mm_scores<NG>(tile, aoff, bq, spart, w, lane, produce);      // dA partial blocks
        mm_barrier();
        float u[4], wgt[4];
        mm_gather(spart, j, kk, u);
        const int jq = j < Q ? j : Q - 1;
        const float* tstat = reinterpret_cast<const float*>(small + C::SMALL);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float tmv = 0.f, trv = 1.f;
          if (LN) {
            tmv = tstat[2 * (4 * s + kk)]; trv = tstat[2 * (4 * s + kk) + 1];
            u[s] = trv * (u[s] - tmv * gsum_j);                                                   // dA = dP . xhat
          }
          const float sv = *reinterpret_cast<const float*>(small + 4 * (16 * jq + 4 * s + kk));   // S[b, j, n0 + 4s + kk]
          const float a = __builtin_amdgcn_exp2f(fmaf(sv, MM_LOG2E, -mL_j)) * il_j;
          wgt[s] = ((4 * s + kk) < nvalid && j < Q) ? a * (u[s] - dl_j) : 0.f;
          if (LN) { wgt[s] *= trv; c3 = fmaf(wgt[s], tmv, c3); }                                  // dS rstd: sum dS xhat
        }
        mm_pool<NG>(tile, plane, w, wgt, gacc);
      }
      if (++cidx == items_per_img) cidx = 0;
    }
  }
  const float gshift = LN ? q4_sum(c3) : 0.f;           // (every wave saw the same weights: no cross-wave term)
  if (j < Q) {
    float* Gq = p.Gpart + ((int64_t)wg * Q + j) * D + 16 * NG * w + 4 * kk;
#pragma unroll
    for (int blk = 0; blk < NG; ++blk) *reinterpret_cast<f4*>(Gq + 16 * blk) = gacc[blk] - gshift;
  }
}

// ---------------------------------------------------------------------------------------
template <int NG, bool LN>
static int mm_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  using C = MmCfg<NG>;
  if constexpr (!C::VALID) {
    set_error("no matrix-core pooling kernel for D=%d", 128 * NG);
    return EP_E_UNSUPPORTED;
  } else {
    const size_t lds = bwd ? (size_t)C::nslot(true, LN) * (C::SLOT + C::SMALL + (LN ? C::TS : 0)) + C::SPART
                           : (size_t)C::nslot(false, LN) * (C::SLOT + (LN ? C::TS : 0)) + C::SPART;
    auto kf = ep_pool_mm_fwd_kernel<NG, LN>;
    auto kb = ep_pool_mm_bwd_kernel<NG, LN>;
    const void* fn = bwd ? (const void*)kb : (const void*)kf;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
    if (bwd) hipLaunchKernelGGL(kb, dim3(grid), dim3(MM_NW * 64), lds, st, p);
    else hipLaunchKernelGGL(kf, dim3(grid), dim3(MM_NW * 64), lds, st, p);
    EP_LAUNCH_CHECK(bwd ? "ep_pool_mm_bwd_kernel" : "ep_pool_mm_fwd_kernel");
    return 0;
  }
}

bool mm_supported(int D, int Q, int64_t cls_bstride) {
  return D % 128 == 0 && D >= 256 && D <= 1152 && Q >= 1 && Q <= 16 && cls_bstride == 0;
}

template <int NG>
static int mm_launch_ln(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  return p.tokstat ? mm_launch_one<NG, true>(bwd, p, grid, st) : mm_launch_one<NG, false>(bwd, p, grid, st);
}

int mm_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  switch (p.D / 128) {
    case 2: return mm_launch_ln<2>(bwd, p, grid, st);
    case 3: return mm_launch_ln<3>(bwd, p, grid, st);
    case 4: return mm_launch_ln<4>(bwd, p, grid, st);
    case 5: return mm_launch_ln<5>(bwd, p, grid, st);
    case 6: return mm_launch_ln<6>(bwd, p, grid, st);
    case 7: return mm_launch_ln<7>(bwd, p, grid, st);
    case 8: return mm_launch_ln<8>(bwd, p, grid, st);
    case 9: return mm_launch_ln<9>(bwd, p, grid, st);
  }
  set_error("no matrix-core pooling kernel for D=%d", p.D);
  return EP_E_UNSUPPORTED;
}

}  // namespace ep
