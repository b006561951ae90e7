// EP attentive pooling, matrix-core variant (gfx950 / CDNA4): the token x query score contraction
// runs on v_mfma_f32_16x16x4_f32 (exact fp32: bit-identical to an fmaf chain), the softmax is
// evaluated lane-parallel and only the attention-weighted reduction stays on the vector ALU.
//
//   forward  (reference poolings/ep.py:35-44):  S = (cls*scale) x^T ; A = softmax_n S ; P = A x
//   backward (autograd of the same lines)     :  dA = dP x^T ; dS = A (dA - delta) ;
//                                                dcls = scale * sum_b dS x
//
// One 8-wave workgroup per CU streams whole images through a ring of 16-token tiles filled by
// LDS-DMA (global_load_lds_dwordx4).  Per tile:
//   1. every wave multiplies the 16 token rows by all (<=16) query rows over ITS eighth of the
//      D axis:  D/32 MFMAs per wave.  The A operand (tokens) is read from the LDS tile with
//      ds_read_b128 -- the tile is stored with its 16-byte chunks XOR-swizzled by the row index
//      (done for free on the DMA source address), which makes this row-strided read and the
//      row-contiguous read of step 3 both bank-conflict free; the B operand (queries) lives in
//      registers.  The DMA instructions that refill the ring are issued in the shadow of the
//      first MFMAs.
//   2. the 8 partial 16x16 score blocks are summed through a small LDS scratch; wave (h, r)
//      (h = w / 4, r = w % 4) ends up with the scores of queries QP*r .. QP*r+QP-1 for the tokens
//      8h .. 8h+7 of the tile -- one (query, token) pair per lane.
//   3. lazy-max online softmax on those lanes, weights broadcast with v_readlane, and the pooled
//      vectors of the wave's queries are accumulated over the FULL D axis with packed FMAs,
//      reading each of its 8 token rows once (row-contiguous ds_read_b128).
//   At the end of an image the two token-halves (h = 0, 1) of every query are merged through LDS.
// Compared with the all-VALU kernel (ep_pool_stream.hip) this issues ~3x fewer vector
// instructions per token, which is what the HBM stream was waiting on.
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"

namespace ep {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int MF_TT = 16;             // tokens per tile = MFMA M
constexpr int MF_HT = 8;              // tokens per wave per tile (token half)
constexpr int MF_NW = 8;              // waves per workgroup (one workgroup per CU)
constexpr float MF_LOG2E = 1.4426950408889634f;
constexpr float MF_LAZY_MAX_THR = 12.0f;

template <int NG>
struct MfCfg {
  static constexpr int D = 128 * NG;
  static constexpr int KP = (NG + 1) / 2;            // 16-byte chunks per lane per row = ceil(D/256)
  static constexpr int ROWB = 4 * D;
  static constexpr int SLOT = MF_TT * ROWB;
  static constexpr int KDMA = NG;                    // 1 KiB DMA pieces per wave per tile (SLOT/1024/8)
  static constexpr int SPART = MF_NW * 272 * 4;      // partial score blocks (padded: conflict-free gather)
  static constexpr int SMALL = MF_NW * 256;          // per slot: 64 floats per wave (backward only)
  static constexpr int LDS_TOTAL = 160 * 1024;
  static constexpr int nslot(bool bwd) {
    int ns = (LDS_TOTAL - SPART) / (SLOT + (bwd ? SMALL : 0));
    return ns > 4 ? 4 : ns;
  }
  static constexpr int NSLOT_F = nslot(false), NSLOT_B = nslot(true);
  static constexpr bool VALID = NSLOT_F >= 2 && NSLOT_B >= 2;
};

__device__ __forceinline__ void mf_wait_vmcnt(int n) {
#define EP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    EP_W(0) EP_W(1) EP_W(2) EP_W(3) EP_W(4) EP_W(5) EP_W(6) EP_W(7) EP_W(8) EP_W(9)
    EP_W(10) EP_W(11) EP_W(12) EP_W(13) EP_W(14) EP_W(15) EP_W(16) EP_W(17) EP_W(18) EP_W(19)
    EP_W(20) EP_W(21) EP_W(22) EP_W(23) EP_W(24) EP_W(25) EP_W(26) EP_W(27) EP_W(28) EP_W(29)
    EP_W(30)
    default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
  }
#undef EP_W
}
template <int N>
__device__ __forceinline__ void mf_wait_vmcnt_imm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate out of range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void mf_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// reductions over the 8 lanes that share a query (lanes 8g .. 8g+7): result in every lane of the group
__device__ __forceinline__ float grp8_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  return v;
}
__device__ __forceinline__ float grp8_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  return v;
}

// per-lane source offsets of this wave's DMA pieces: LDS position (piece pc, lane) holds chunk
// (c ^ (t & 15)) of row t, where (t, c) = divmod(pc*64 + lane, chunks per row).
template <int NG>
__device__ __forceinline__ void mf_source_offsets(int w, int lane, unsigned (&soff)[NG]) {
  constexpr int NCHUNK = 32 * NG;
#pragma unroll
  for (int jj = 0; jj < NG; ++jj) {
    const int pos = (w + MF_NW * jj) * 64 + lane;
    const int t = pos / NCHUNK, c = pos - t * NCHUNK;
    soff[jj] = (unsigned)(t * (512 * NG) + ((c ^ (t & 15)) << 4));
  }
}
template <int NG>
__device__ __forceinline__ void mf_dma_tile(const char* src, unsigned limit, char* slot, int w,
                                            const unsigned (&soff)[NG]) {
#pragma unroll
  for (int jj = 0; jj < NG; ++jj) {
    const unsigned off = soff[jj] < limit ? soff[jj] : limit;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lds_ptr_t)(slot + (w + MF_NW * jj) * 1024), 16, 0,
                                     EP_DMA_AUX);
  }
}

// 16 tokens x 16 queries over this wave's D-slice -> LDS scratch.  `mid` runs after the first
// MFMAs have been issued (the ring refill goes there, in the shadow of the matrix pipe).
template <int NG, typename F>
__device__ __forceinline__ void mf_scores(const char* tile, const int (&aoff)[NG], const float (&bq)[NG][4],
                                          char* spart, int w, int lane, F&& mid) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 xa[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) xa[g] = *reinterpret_cast<const f4*>(tile + aoff[g]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].x, bq[g][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].y, bq[g][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].z, bq[g][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].w, bq[g][3], acc, 0, 0, 0);
    if (g == 0) mid();
  }
  *reinterpret_cast<f4*>(spart + (w * 272 + lane * 4 + (lane >> 4) * 4) * 4) = acc;
}
// score of (query q, token t): D layout col = query, row = token -> lane 16*(t>>2)+q, reg t&3
__device__ __forceinline__ float mf_gather(const char* spart, int q, int t) {
  const float* base = reinterpret_cast<const float*>(spart) + (16 * (t >> 2) + q) * 4 + (t >> 2) * 4 + (t & 3);
  float s = 0.f;
#pragma unroll
  for (int ws = 0; ws < MF_NW; ++ws) s += base[ws * 272];
  return s;
}

// acc[qi][k] += sum_{tt < rows} wlane(8*qi + tt) * x[t0 + tt, chunk k of this lane]
// (rows are read through the XOR swizzle; every row is read once and used for all QP queries)
template <int NG, int QP>
__device__ __forceinline__ void mf_pool(const char* tile, int t0, int rows, float wlane, unsigned lane16,
                                        f4 (&acc)[QP][(NG + 1) / 2]) {
  constexpr int KP = (NG + 1) / 2, ROWB = 512 * NG;
  auto load_row = [&](int t, f4 (&dst)[KP]) {
    const char* rowp = tile + t * ROWB + (lane16 ^ (unsigned)((t & 15) << 4));
#pragma unroll
    for (int k = 0; k < KP; ++k) dst[k] = *reinterpret_cast<const f4*>(rowp + 1024 * k);
  };
  if (rows == MF_HT) {
    // full half-tile: two groups of 4 rows; the second group's reads are issued before the first
    // group's FMAs so the LDS latency is covered
    f4 xv[2][4][KP];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_row(t0 + u, xv[0][u]);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) load_row(t0 + 4 + u, xv[1][u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) {
          const float a = readlane_f(wlane, 8 * qi + 4 * g + u);
#pragma unroll
          for (int k = 0; k < KP; ++k) acc[qi][k] += a * xv[g][u][k];
        }
    }
  } else {                                   // last tile of an image
    for (int tt = 0; tt < rows; ++tt) {
      f4 xr[KP];
      load_row(t0 + tt, xr);
#pragma unroll
      for (int qi = 0; qi < QP; ++qi) {
        const float a = readlane_f(wlane, 8 * qi + tt);
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[qi][k] += a * xr[k];
      }
    }
  }
}

__device__ __forceinline__ unsigned long long mf_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define MF_STAMP(slot)                                          \
  if constexpr (STAMP) {                                        \
    const unsigned long long now__ = mf_stamp();                \
    stamp_acc[slot] += now__ - stamp_last;                      \
    stamp_last = now__;                                         \
  }

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int NG, int QP, bool STAMP = false>
__global__ __launch_bounds__(MF_NW * 64, 2) void ep_pool_mf_fwd_kernel(PoolParams p) {
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = 0;
  if constexpr (STAMP) stamp_last = mf_stamp();
  using C = MfCfg<NG>;
  constexpr int D = C::D, KP = C::KP, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT_F, KDMA = C::KDMA;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const int tiles_per_img = (N + MF_TT - 1) / MF_TT;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  const int j = lane & 15, kk = lane >> 4;
  const unsigned lane16 = (unsigned)lane * 16u;
  // pooling role of this wave: token half h, queries qbase .. qbase+QP-1; lane = (query qi, token tt)
  const int h = w >> 2, r = w & 3, qbase = QP * r;
  const int qi_l = (lane >> 3) & (QP - 1), tt_l = lane & 7;
  const int q_l = (qbase + qi_l) < Q ? (qbase + qi_l) : Q - 1;      // clamped: padded queries mirror the last one
  const int t_l = MF_HT * h + tt_l;

  // B operand: queries pre-scaled like the reference (q = cls_token * scale, ep.py:39); lane (j,kk)
  // holds query j, k = 16*NG*w + 16g + 4kk + s
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
  unsigned soff[NG];
  mf_source_offsets<NG>(w, lane, soff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, wg));
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * MF_TT;
      const unsigned limit = (unsigned)((left < MF_TT ? left : MF_TT) * ROWB - 16);
      mf_dma_tile<NG>(psrc, limit, ring + pslot * SLOT, w, soff);
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        psrc = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg));
      } else {
        psrc += SLOT;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 acc[QP][KP];
  float m_l = -INFINITY, mL_l = -INFINITY, lsum_l = 0.f;   // per lane: state of query qi_l
  int cimg = 0, ctile = 0, cslot = 0;
  for (int i = 0; i < n_items; ++i) {
    // NOTE on the counted wait: `produce` for this iteration runs inside mf_scores, i.e. AFTER this
    // wait, so the DMA ops outstanding here are those of items i .. pi-1 (KDMA each) plus stores.
    const int ahead = pi - 1 - i;
    MF_STAMP(0)
    if (ahead == NSLOT - 2) mf_wait_vmcnt_imm<(NSLOT - 2) * KDMA>();
    else mf_wait_vmcnt(ahead * KDMA);
    MF_STAMP(1)
    mf_barrier();                                   // tile i landed everywhere; slot of tile i-1 is free
    MF_STAMP(2)
    const int b = wg + cimg * G;
    const int n0 = ctile * MF_TT;
    const int nvalid = (N - n0) < MF_TT ? (N - n0) : MF_TT;
    const char* tile = ring + cslot * SLOT;
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    if (ctile == 0) {
      m_l = -INFINITY; mL_l = -INFINITY; lsum_l = 0.f;
#pragma unroll
      for (int qi = 0; qi < QP; ++qi)
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[qi][k] = f4{0.f, 0.f, 0.f, 0.f};
    }
    mf_scores<NG>(tile, aoff, bq, spart, w, lane, produce);
    MF_STAMP(4)
    mf_barrier();                                   // all partial score blocks are in the scratch
    MF_STAMP(5)
    int rows = nvalid - MF_HT * h;
    rows = rows < 0 ? 0 : (rows > MF_HT ? MF_HT : rows);
    if (qbase < Q && rows > 0) {                    // wave-uniform
      const float s = mf_gather(spart, q_l, t_l);
      const bool valid = tt_l < rows;
      const float ue = valid ? s : -INFINITY;
      if (__builtin_amdgcn_ballot_w64(ue > m_l + MF_LAZY_MAX_THR) != 0ull) {   // rare
        const float mn = fmaxf(m_l, grp8_max(ue));
        const float f = __builtin_amdgcn_exp2f((m_l - mn) * MF_LOG2E);        // m = -inf -> 0
        m_l = mn; mL_l = mn * MF_LOG2E;
        lsum_l *= f;
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) {
          const float fq = readlane_f(f, 8 * qi);
#pragma unroll
          for (int k = 0; k < KP; ++k) acc[qi][k] *= fq;
        }
      }
      const float pr = __builtin_amdgcn_exp2f(fmaf(ue, MF_LOG2E, -mL_l));
      lsum_l += pr;
      if (lane < 8 * QP && valid && (qbase + qi_l) < Q)
        p.S[((int64_t)b * QS + q_l) * N + (unsigned)(n0 + t_l)] = s;
      MF_STAMP(6)
      mf_pool<NG, QP>(tile, MF_HT * h, rows, pr, lane16, acc);
      MF_STAMP(7)
    }
    if (ctile == tiles_per_img - 1) {
      // ---- image epilogue: merge the two token halves of every query through the (now idle)
      //      tile slot, normalise, store.  The slot is refilled only after the next ring barrier.
      char* scratch = const_cast<char*>(tile);
      float* hdr = reinterpret_cast<float*>(spart);           // [4 waves][QP][2]: m, l of the h = 1 halves
      const float lq = grp8_sum(lsum_l);
      mf_barrier();                                           // everyone is done reading the tile / scratch
      if (h == 1) {
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) {
#pragma unroll
          for (int k = 0; k < KP; ++k)
            *reinterpret_cast<f4*>(scratch + ((r * QP + qi) * KP + k) * 1024 + lane16) = acc[qi][k];
          if (lane == 8 * qi) { hdr[(r * QP + qi) * 2 + 0] = m_l; hdr[(r * QP + qi) * 2 + 1] = lq; }
        }
      }
      mf_barrier();
      if (h == 0) {
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) {
          const int q = qbase + qi;
          if (q < Q) {
            const float m1 = readlane_f(m_l, 8 * qi), l1 = readlane_f(lq, 8 * qi);
            const float m2 = hdr[(r * QP + qi) * 2 + 0], l2 = hdr[(r * QP + qi) * 2 + 1];
            const float mn = fmaxf(m1, m2);
            const float f1 = __builtin_amdgcn_exp2f((m1 - mn) * MF_LOG2E);
            const float f2 = __builtin_amdgcn_exp2f((m2 - mn) * MF_LOG2E);
            const float l = l1 * f1 + l2 * f2;
            const float i1 = f1 / l, i2 = f2 / l;
            float* Pq = p.P + ((int64_t)b * QS + q) * D;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
              const f4 other = *reinterpret_cast<const f4*>(scratch + ((r * QP + qi) * KP + k) * 1024 + lane16);
              const int c = lane + 64 * k;
              if (c < D / 4) *reinterpret_cast<f4*>(Pq + 4 * c) = acc[qi][k] * i1 + other * i2;
            }
            if (lane == 0) {
              const f4 rec = {mn, l, 0.f, 0.f};
              *reinterpret_cast<f4*>(p.ML + ((int64_t)b * QS + q) * 4) = rec;
            }
          }
        }
      }
      ctile = 0; ++cimg;
    } else {
      ++ctile;
    }
  }
  if constexpr (STAMP) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p.dbg[((int64_t)wg * MF_NW + w) * 8 + k] = stamp_acc[k];
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward.  Ring items per image: one header tile holding the Q rows of dP[b] (the B operand of
// the image), then the token tiles.  Every item carries one 4-byte-per-lane DMA per wave into a
// private 256-byte area: the header brings ML[b, q, 0:4] of the wave's queries, token tiles bring
// S[b, q, n0+8h : n0+8h+8].  Each wave writes its own partial of the cls_token gradient
// (2 * gridDim.x partials: token halves are summed by ep_reduce_partials).
// ---------------------------------------------------------------------------------------
template <int NG, int QP>
__global__ __launch_bounds__(MF_NW * 64, 2) void ep_pool_mf_bwd_kernel(PoolParams p) {
  using C = MfCfg<NG>;
  constexpr int D = C::D, KP = C::KP, ROWB = C::ROWB, SLOT = C::SLOT, NSLOT = C::NSLOT_B, KDMA = C::KDMA;
  constexpr int KD = KDMA + 1;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* ring = lds;
  char* spart = lds + NSLOT * SLOT;
  char* small_base = spart + C::SPART;                  // [NSLOT][8 waves][64 floats]
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int N = p.N, Q = p.Q;
  const int QS = p.Qs ? p.Qs : p.Q;          // queries per image in memory (a launch may cover a chunk of them)
  const int tiles_per_img = (N + MF_TT - 1) / MF_TT;
  const int items_per_img = 1 + tiles_per_img;
  const int G = gridDim.x, wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * items_per_img;
  const int j = lane & 15, kk = lane >> 4;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int h = w >> 2, r = w & 3, qbase = QP * r;
  const int qi_l = (lane >> 3) & (QP - 1), tt_l = lane & 7;
  const int q_l = (qbase + qi_l) < Q ? (qbase + qi_l) : Q - 1;
  const int t_l = MF_HT * h + tt_l;

  f4 gacc[QP][KP];
#pragma unroll
  for (int qi = 0; qi < QP; ++qi)
#pragma unroll
    for (int k = 0; k < KP; ++k) gacc[qi][k] = f4{0.f, 0.f, 0.f, 0.f};

  if (n_items > 0) {
    int aoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) aoff[g] = j * ROWB + (((4 * NG * w + 4 * g + kk) ^ j) << 4);
    unsigned soff[NG];
    mf_source_offsets<NG>(w, lane, soff);
    // lane -> element of the small DMA (lanes 0 .. 8*QP-1 carry data, the rest mirror)
    const int hq = (qbase + ((lane >> 2) & (QP - 1))) < Q ? (qbase + ((lane >> 2) & (QP - 1))) : Q - 1;

    int pi = 0, pimg = 0, pidx = 0, pslot = 0;
    auto produce = [&]() {
      if (pi < n_items) {
        const int b = wg + pimg * G;
        char* slot = ring + pslot * SLOT;
        char* small = small_base + (pslot * MF_NW + w) * 256;
        if (pidx == 0) {
          const char* src = reinterpret_cast<const char*>(p.dP + (int64_t)b * QS * D);
          const int rows = Q < MF_TT ? Q : MF_TT;
          mf_dma_tile<NG>(src, (unsigned)(rows * ROWB - 16), slot, w, soff);
          const float* ms = p.ML + ((int64_t)b * QS + hq) * 4 + (lane & 3);      // lane = 4*qi + field
          __builtin_amdgcn_global_load_lds((gptr_t)ms, (lds_ptr_t)small, 4, 0, 0);
        } else {
          const int n0 = (pidx - 1) * MF_TT;
          const int rows = (N - n0) < MF_TT ? (N - n0) : MF_TT;
          const char* src = reinterpret_cast<const char*>(p.x + EP_IMG_OFF(p, b) + (int64_t)n0 * D);
          mf_dma_tile<NG>(src, (unsigned)(rows * ROWB - 16), slot, w, soff);
          int nn = n0 + t_l; nn = nn < N ? nn : N - 1;
          const float* ss = p.S + ((int64_t)b * QS + q_l) * N + nn;              // lane = 8*qi + tt
          __builtin_amdgcn_global_load_lds((gptr_t)ss, (lds_ptr_t)small, 4, 0, 0);
        }
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++pidx == items_per_img) { pidx = 0; ++pimg; }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    float bq[NG][4];
    float mL_l = 0.f, il_l = 0.f, dl_l = 0.f;       // per lane: row max*log2e, 1/l, delta of query qi_l
    int cidx = 0, cslot = 0;
    for (int i = 0; i < n_items; ++i) {
      const int ahead = pi - 1 - i;
      if (ahead == NSLOT - 2) mf_wait_vmcnt_imm<(NSLOT - 2) * KD>();
      else mf_wait_vmcnt(ahead * KD);
      mf_barrier();
      const char* tile = ring + cslot * SLOT;
      const float* small = reinterpret_cast<const float*>(small_base + (cslot * MF_NW + w) * 256);
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      if (cidx == 0) {
        produce();
        // header: this image's B operand (rows of dP, same swizzled image as a token tile)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          f4 v = *reinterpret_cast<const f4*>(tile + aoff[g]);
          if (j >= Q) v = f4{0.f, 0.f, 0.f, 0.f};
          bq[g][0] = v.x; bq[g][1] = v.y; bq[g][2] = v.z; bq[g][3] = v.w;
        }
        mL_l = small[4 * qi_l + 0] * MF_LOG2E;
        il_l = 1.0f / small[4 * qi_l + 1];
        dl_l = small[4 * qi_l + 2];
      } else {
        const int n0 = (cidx - 1) * MF_TT;
        const int nvalid = (N - n0) < MF_TT ? (N - n0) : MF_TT;
        mf_scores<NG>(tile, aoff, bq, spart, w, lane, produce);      // dA partial blocks
        mf_barrier();
        int rows = nvalid - MF_HT * h;
        rows = rows < 0 ? 0 : (rows > MF_HT ? MF_HT : rows);
        if (qbase < Q && rows > 0) {
          const float u = mf_gather(spart, q_l, t_l);
          const float s = small[8 * qi_l + tt_l];
          const float a = __builtin_amdgcn_exp2f(fmaf(s, MF_LOG2E, -mL_l)) * il_l;
          const float wgt = (tt_l < rows) ? a * (u - dl_l) : 0.f;
          mf_pool<NG, QP>(tile, MF_HT * h, rows, wgt, lane16, gacc);
        }
      }
      if (++cidx == items_per_img) cidx = 0;
    }
  }
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    const int q = qbase + qi;
    if (q < Q) {
      float* Gq = p.Gpart + (((int64_t)wg * 2 + h) * Q + q) * D;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int c = lane + 64 * k;
        if (c < D / 4) *reinterpret_cast<f4*>(Gq + 4 * c) = gacc[qi][k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------
// diagnostic only (EP_MF_STAMP=1): per-phase cycle shares of the forward kernel, printed to stderr
static int mf_launch_stamped(const PoolParams& p0, int grid, size_t lds, hipStream_t st) {
  PoolParams p = p0;
  static unsigned long long* dbg = nullptr;
  const size_t n = (size_t)grid * MF_NW * 8;
  if (!dbg) (void)hipMalloc(&dbg, 4096 * 8 * sizeof(unsigned long long));
  (void)hipMemsetAsync(dbg, 0, n * sizeof(unsigned long long), st);
  p.dbg = dbg;
  auto k = ep_pool_mf_fwd_kernel<6, 2, true>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(MF_NW * 64), lds, st, p);
  (void)hipStreamSynchronize(st);
  static unsigned long long host[4096 * 8];
  (void)hipMemcpy(host, dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double tot[8] = {0};
  for (size_t i = 0; i < n; ++i) tot[i % 8] += (double)host[i];
  double all = 0;
  for (int k2 = 0; k2 < 8; ++k2) all += tot[k2];
  const char* names[8] = {"bookkeeping", "dma-wait", "ring-barrier", "(unused)", "score-mfma+dma-issue",
                          "score-barrier", "gather+softmax", "pool"};
  fprintf(stderr, "[EP_MF_STAMP] cycles per wave (avg over %d waves): total %.0f\n", grid * MF_NW,
          all / (grid * MF_NW));
  for (int k2 = 0; k2 < 8; ++k2)
    fprintf(stderr, "   %-22s %10.0f  %5.1f%%\n", names[k2], tot[k2] / (grid * MF_NW), 100.0 * tot[k2] / all);
  return 0;
}

template <int NG, int QP>
static int mf_launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  using C = MfCfg<NG>;
  if constexpr (!C::VALID) {
    set_error("no matrix-core pooling kernel for D=%d", 128 * NG);
    return EP_E_UNSUPPORTED;
  } else {
    const size_t lds = bwd ? (size_t)C::NSLOT_B * (C::SLOT + C::SMALL) + C::SPART
                           : (size_t)C::NSLOT_F * C::SLOT + C::SPART;
    if constexpr (NG == 6 && QP == 2) {
      if (!bwd && getenv("EP_MF_STAMP")) return mf_launch_stamped(p, grid, lds, st);
    }
    auto kf = ep_pool_mf_fwd_kernel<NG, QP>;
    auto kb = ep_pool_mf_bwd_kernel<NG, QP>;
    const void* fn = bwd ? (const void*)kb : (const void*)kf;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
    if (bwd) hipLaunchKernelGGL(kb, dim3(grid), dim3(MF_NW * 64), lds, st, p);
    else hipLaunchKernelGGL(kf, dim3(grid), dim3(MF_NW * 64), lds, st, p);
    EP_LAUNCH_CHECK(bwd ? "ep_pool_mf_bwd_kernel" : "ep_pool_mf_fwd_kernel");
    return 0;
  }
}

template <int QP>
static int mf_dispatch_ng(bool bwd, int ng, const PoolParams& p, int grid, hipStream_t st) {
  switch (ng) {
    case 2: return mf_launch_one<2, QP>(bwd, p, grid, st);
    case 3: return mf_launch_one<3, QP>(bwd, p, grid, st);
    case 4: return mf_launch_one<4, QP>(bwd, p, grid, st);
    case 5: return mf_launch_one<5, QP>(bwd, p, grid, st);
    case 6: return mf_launch_one<6, QP>(bwd, p, grid, st);
    case 7: return mf_launch_one<7, QP>(bwd, p, grid, st);
    case 8: return mf_launch_one<8, QP>(bwd, p, grid, st);
    case 9: return mf_launch_one<9, QP>(bwd, p, grid, st);
  }
  set_error("no matrix-core pooling kernel for D=%d", 128 * ng);
  return EP_E_UNSUPPORTED;
}

bool mf_supported(int D, int Q, int64_t cls_bstride) {
  // The forward's image epilogue merges the two token halves through the idle tile slot: 4 * QP * KP KiB of the slot's
  // 64 * D bytes.  With QP = 4 (Q > 8) that only fits when D / 128 is even (KP = ceil(D / 256)): odd widths such as
  // 1152 overran the slot into the next tile, so they take another kernel family.
  if (Q > 8 && (D / 128) % 2 != 0) return false;
  return D % 128 == 0 && D >= 256 && D <= 1152 && Q >= 1 && Q <= 16 && cls_bstride == 0;
}
int mf_partials_per_wg() { return 2; }

int mf_launch(bool bwd, const PoolParams& p, int grid, hipStream_t st) {
  const int ng = p.D / 128;
  return p.Q > 8 ? mf_dispatch_ng<4>(bwd, ng, p, grid, st) : mf_dispatch_ng<2>(bwd, ng, p, grid, st);
}

}  // namespace ep
