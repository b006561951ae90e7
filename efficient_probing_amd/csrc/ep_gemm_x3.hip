// fp32 tensor contractions on the gfx950 BF16 matrix cores, at fp32 accuracy (three-term operand split).
//
//   C[z][m][n] (+)= alpha * sum_k A[z](m,k) * B[z](k,n)  (+ bias[n])           (same contract as ep_gemm.hip)
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD), 1/16 of the bf16 matrix rate.  The
// head's 1024-row contractions (value projection y = P Wv_q^T, classifier logits, dz, dP -- reference poolings/ep.py:40,
// probe_heads.py:76 and their autograd) are one 64x64 tile per CU, i.e. ~10 us of f32 matrix time each whatever the
// schedule.  Every fp32 value is the EXACT sum of three bf16 values (8 + 8 + 8 significant bits, split by truncation):
//   x = h + m + l,   h = trunc16(x),  m = trunc16(x - h),  l = x - h - m            (both subtractions are exact)
// and a bf16 x bf16 product is exact in fp32, so
//   a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [am*bl + al*bm + al*bl]
// where the bracket is <= 2^-23 |a*b| (dropped: the size of one fp32 rounding).  Six v_mfma_f32_16x16x32_bf16 per
// 16x16x32 block therefore replace eight v_mfma_f32_16x16x4_f32 at 6 x 16 = 96 instead of 8 x 32 = 256 matrix cycles,
// with fp32 accumulation and fewer accumulator roundings than the fmaf chain of the f32 instruction (K/32 * 6 against K).
//
// Structure: the proven LDS-DMA ring of ep_gemm_dma_kernel (fp32 operand tiles HBM/L2 -> LDS by global_load_lds_dwordx4,
// XOR-swizzled on the source address, counted vmcnt, one barrier per K-tile) is kept unchanged; the split happens on the
// FRAGMENTS, in registers, after the conflict-free fragment reads -- no second LDS image, no extra pass over the
// operands.  The 32 k-values of a K-tile map onto ONE bf16 MFMA: element e = 4g + j of lane group kk holds
// k = 16g + 4kk + j for BOTH operands (any bijection is valid for the contraction), which is exactly what the fp32
// fragment reads deliver (one ds_read_b128 per g for K-layout operands, four ds_read_b32 for T-layout ones).
// The split costs ~5.5 vector instructions per fragment element, so the wave tile is chosen to balance vector issue and
// matrix time: 8 symmetric waves (two per SIMD), each 32 rows x 16 columns of the 64x64 workgroup tile -- per K-tile a
// wave splits 24 elements per lane (~130 vector instructions) for 12 MFMAs, and its SIMD partner's MFMAs run under them.
#include "ep_side.h"

namespace ep {

typedef __attribute__((address_space(3))) void* x3_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* x3_gptr_t;
typedef __bf16 x3_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned x3_u4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void x3_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void x3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// two fp32 values -> their three bf16 terms, packed (element 0 in the low half of each register)
__device__ __forceinline__ void x3_split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
  const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);          // exact
  const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
  const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);          // exact, <= 8 significant bits
  h = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  m = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
// the 8 fragment elements of one 16-row block (g = 0, 1; j = 0..3) -> three bf16x8 operands
__device__ __forceinline__ void x3_split8(const f4v (&x)[2], x3_u4 (&t)[3]) {
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      unsigned h, m, l;
      x3_split2(x[g][2 * q], x[g][2 * q + 1], h, m, l);
      t[0][2 * g + q] = h; t[1][2 * g + q] = m; t[2][2 * g + q] = l;
    }
}
__device__ __forceinline__ f4v x3_mfma(x3_u4 a, x3_u4 b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3_bf8, a), __builtin_bit_cast(x3_bf8, b), c, 0, 0, 0);
}

template <bool A_K, bool B_K, int NST>
__global__ __launch_bounds__(512) void ep_gemm_x3_kernel(GemmParams p) {
  constexpr int OPB = 64 * BK * 4;                   // bytes per fp32 operand image (8 KiB)
  constexpr int STB = 2 * OPB;                       // bytes per ring stage
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // NST * STB bytes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
  const int wm = w >> 2, wn = w & 3;                 // wave tile: rows 32 wm .. +31, columns 16 wn .. +15
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int nk = (p.K + BK - 1) / BK;

  // ---- DMA: 8 pieces of 1 KiB per operand image; wave w moves piece w of A and piece w of B (the same swizzled images
  // as ep_gemm_dma_kernel: K layout chunk c of row r in slot c ^ ((r >> 1) & 7), T layout chunk c of k-row k in slot
  // c ^ (4 ((k >> 2) & 1)))
  auto src_off = [&](bool klay, int64_t ld, int lim, int ext, int r0, int pos, int k0) -> int64_t {
    if (klay) {
      const int r = pos >> 3, q = pos & 7;
      const int kq = q ^ ((r >> 1) & 7);
      int row = r0 + r; row = row < lim ? row : lim - 1;
      int k = k0 + 4 * kq; k = k < p.K ? k : 0;
      return (int64_t)row * ld + k;
    } else {
      const int k = pos >> 4, q = pos & 15;
      const int c = q ^ (4 * ((k >> 2) & 1));
      int kr = k0 + k; kr = kr < p.K ? kr : p.K - 1;
      int col = r0 + 4 * c; col = col < ext ? col : r0;
      return (int64_t)kr * ld + col;
    }
  };
  const int pos = w * 64 + lane;
  const int64_t srcA = src_off(A_K, p.lda, p.M, p.extA, m0, pos, 0);
  const int64_t srcB = src_off(B_K, p.ldb, p.N, p.extB, n0, pos, 0);
  const int64_t kstepA = A_K ? BK : (int64_t)BK * p.lda;
  const int64_t kstepB = B_K ? BK : (int64_t)BK * p.ldb;
  const bool ktail = (p.K % BK) != 0;
  auto issue = [&](int t) {                          // DMA K-tile t (clamped to the last one) into stage t % NST
    const int tt = t < nk ? t : nk - 1;
    char* st = lds + (t % NST) * STB;
    int64_t oa, ob;
    if (ktail && tt == nk - 1) {
      oa = src_off(A_K, p.lda, p.M, p.extA, m0, pos, tt * BK);
      ob = src_off(B_K, p.ldb, p.N, p.extB, n0, pos, tt * BK);
    } else {
      oa = srcA + tt * kstepA; ob = srcB + tt * kstepB;
    }
    __builtin_amdgcn_global_load_lds((x3_gptr_t)(A + oa), (x3_lds_ptr_t)(st + w * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((x3_gptr_t)(B + ob), (x3_lds_ptr_t)(st + OPB + w * 1024), 16, 0, 0);
  };

  // ---- fragment addressing (bytes inside an operand image), as ep_gemm_dma_kernel ----
  int fragA[2][2], fragB[2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
    const int r = wm * 32 + bi * 16 + i16;
    if (A_K) {
      fragA[bi][0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
      fragA[bi][1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
    } else {
      fragA[bi][0] = 4 * kk * 256 + 16 * ((r >> 2) ^ (4 * (kk & 1))) + 4 * (r & 3);
      fragA[bi][1] = 0;
    }
  }
  {
    const int r = wn * 16 + i16;
    if (B_K) {
      fragB[0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
      fragB[1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
    } else {
      fragB[0] = 4 * kk * 256 + 16 * ((r >> 2) ^ (4 * (kk & 1))) + 4 * (r & 3);
      fragB[1] = 0;
    }
  }
  f4v fa[2][2][2], fb[2][2];                         // [set][block][g], [set][g]
  auto read_frags = [&](int stage, f4v (&xa)[2][2], f4v (&xb)[2]) {
    const char* sa = lds + stage * STB;
    const char* sb = sa + OPB;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        if (A_K) xa[bi][g] = *reinterpret_cast<const f4v*>(sa + fragA[bi][g]);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) xa[bi][g][j] = *reinterpret_cast<const float*>(sa + fragA[bi][0] + (16 * g + j) * 256);
        }
      }
      if (B_K) xb[g] = *reinterpret_cast<const f4v*>(sb + fragB[g]);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[g][j] = *reinterpret_cast<const float*>(sb + fragB[0] + (16 * g + j) * 256);
      }
    }
  };
  f4v acc[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
  auto multiply = [&](const f4v (&xa)[2][2], const f4v (&xb)[2]) {
    x3_u4 b3[3], a3[2][3];
    x3_split8(xb, b3);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) x3_split8(xa[bi], a3[bi]);
    // smallest terms first: lo x hi, hi x lo, mid x mid, then the 2^-8 pair, then hi x hi
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = x3_mfma(a3[bi][2], b3[0], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = x3_mfma(a3[bi][0], b3[2], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = x3_mfma(a3[bi][1], b3[1], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = x3_mfma(a3[bi][1], b3[0], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = x3_mfma(a3[bi][0], b3[1], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = x3_mfma(a3[bi][0], b3[0], acc[bi]);
  };
  auto zero_tail = [&](int k0, f4v (&xa)[2][2], f4v (&xb)[2]) {   // last tile only: k >= K contributes nothing
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool out = k0 + 16 * g + 4 * kk + j >= p.K;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) xa[bi][g][j] = out ? 0.f : xa[bi][g][j];
        xb[g][j] = out ? 0.f : xb[g][j];
      }
  };

  // ---- pipeline (two DMA instructions per wave per K-tile) ----
#pragma unroll
  for (int t = 0; t < NST - 1; ++t) issue(t);
  x3_dma_wait<2 * (NST - 2)>();                      // this wave's pieces of tile 0
  x3_barrier();                                      // ... and every other wave's
  read_frags(0, fa[0], fb[0]);
  if (ktail && nk == 1) { __builtin_amdgcn_s_waitcnt(0xc07f); zero_tail(0, fa[0], fb[0]); }
  // step it (set F = it % 2): tile it+1 landed (vmcnt + barrier; the barrier also says every wave holds tile `it` in
  // registers, so its stage can be refilled) -> DMA tile it+NST-1 into that stage, read the fragments of tile it+1
  // into the other set, split + multiply tile it.
#define EP_X3_STEP(IT, F)                                                          \
  {                                                                                \
    x3_dma_wait<2 * (NST - 3 >= 0 ? NST - 3 : 0)>();                               \
    x3_barrier();                                                                  \
    if (p.ablate != 2) issue((IT) + NST - 1);                                      \
    read_frags(((IT) + 1) % NST, fa[(F) ^ 1], fb[(F) ^ 1]);                        \
    if (p.ablate != 1) multiply(fa[F], fb[F]);                                     \
    if (ktail && (IT) + 1 == nk - 1) { __builtin_amdgcn_s_waitcnt(0xc07f); zero_tail(((IT) + 1) * BK, fa[(F) ^ 1], fb[(F) ^ 1]); } \
  }
  int it = 0;
  for (; it + 1 < nk; it += 2) {
    EP_X3_STEP(it, 0)
    EP_X3_STEP(it + 1, 1)
  }
  if (it < nk) EP_X3_STEP(it, 0)
#undef EP_X3_STEP
  x3_dma_wait<0>();                                  // redundant prefetches past the last tile: drain before exit

  const int col = n0 + wn * 16 + i16;
  if (col < p.N) {
    const float bv = p.bias ? p.bias[(int64_t)z * p.sBiasz + col] : 0.f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + bi * 16 + kk * 4 + r;
        if (row < p.M) {
          float* c = C + (int64_t)row * p.ldc + col;
          float v = p.alpha * acc[bi][r] + bv;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
  }
}

template <bool A_K, bool B_K, int NST>
static void x3_launch_one(const GemmParams& p, dim3 grid, hipStream_t st) {
  constexpr int lds = NST * 2 * 64 * BK * 4;
  auto k = ep_gemm_x3_kernel<A_K, B_K, NST>;
  static bool attr_set = false;                      // per instantiation: raise the dynamic-LDS limit once
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
}

// Ring depth: a contraction with at most one 64x64 tile per CU (the 1024-row head contractions) has the whole LDS of its
// CU and nothing else to hide the operand latency (L2 / Infinity Cache -> LDS): 8 stages = 7 K-tiles (112 KiB) in
// flight.  Larger grids run two workgroups per CU on 4-stage rings (64 KiB each).
void gemm_launch_x3(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + 63) / 64, batch);
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("EP_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
  GemmParams pa = p; pa.ablate = ablate;
  static int force_nst = -1;
  if (force_nst < 0) { const char* e = getenv("EP_GEMM_X3_NST"); force_nst = e ? atoi(e) : 0; }
  const long tiles = (long)grid.x * grid.y * grid.z;
  const bool deep = force_nst ? force_nst == 8 : tiles <= (long)cu_count();
#define EP_GEMM_LAUNCH(AK, BK_) { if (deep) x3_launch_one<AK, BK_, 8>(pa, grid, st); else x3_launch_one<AK, BK_, 4>(pa, grid, st); }
  if (a_k && b_k) EP_GEMM_LAUNCH(true, true)
  else if (a_k && !b_k) EP_GEMM_LAUNCH(true, false)
  else if (!a_k && b_k) EP_GEMM_LAUNCH(false, true)
  else EP_GEMM_LAUNCH(false, false)
#undef EP_GEMM_LAUNCH
}

}  // namespace ep
