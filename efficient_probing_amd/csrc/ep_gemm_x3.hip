// EXPERIMENT (opt-in: EP_GEMM_X3=1; gemm() uses ep_gemm_dma_kernel by default) -- fp32 tensor contractions on the gfx950
// BF16 matrix cores at fp32 accuracy (three-term operand split).
//
//   C[z][m][n] (+)= alpha * sum_k A[z](m,k) * B[z](k,n)  (+ bias[n])           (same contract as ep_gemm.hip)
//
// Idea: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD), 1/16 of the bf16 matrix rate, and the f32
// contraction kernel is bound by it (0.64 us per 64x64x32 K-tile = 1024 matrix cycles at the ~1.65 GHz the chip holds
// under this load; the operand ring alone, no arithmetic, runs at 0.22 us per tile).  Every fp32 value is the EXACT sum of
// three bf16 values (8 + 8 + 8 significant bits, each term rounded to nearest even):
//   x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = x - h - m                  (both subtractions are exact)
// and a bf16 x bf16 product is exact in fp32, so
//   a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [am*bl + al*bm + al*bl]
// where the bracket is <= 2^-24 |a*b| and, with round-to-nearest terms, of either sign (dropped).  Six
// v_mfma_f32_16x16x32_bf16 per 16x16x32 block replace eight v_mfma_f32_16x16x4_f32 at 96 instead of 256 matrix cycles.
// Accuracy: tools/gemm_fuzz.py -- relative error 1.0e-6 at K = 768 against 1.2e-6 for the f32 instruction's fmaf chain.
//
// Structure: every operand element of a K-tile is split ONCE per workgroup (the split costs ~4.5 vector instructions per
// element; a first version that split fragments in registers, redundantly in every wave that multiplies them, was bound
// by that):
//   * the fp32 operand tiles arrive in LDS through the LDS-DMA ring of ep_gemm_dma_kernel (global_load_lds_dwordx4,
//     XOR-swizzled on the source address, counted vmcnt);
//   * per K-tile the waves each take ONE (operand, 16-row block) unit: read its fp32 fragment in the MFMA lane layout,
//     split it, and write the three bf16x8 operands back to LDS in "MFMA-native" order -- plane image [unit][term][lane]
//     with 16 bytes per lane, i.e. a wave writes and later reads 1 KiB linearly (conflict-free by construction);
//   * then each wave multiplies its part of the tile from those images (linear ds_read_b128, 6 MFMAs per block pair).
// The 32 k-values of a K-tile map onto ONE bf16 MFMA: element e = 4g + j of lane group kk holds k = 16g + 4kk + j for BOTH
// operands (any bijection is valid for the contraction).  Split of tile it+1 and multiplication of tile it share one
// barrier interval (two plane buffers): one s_barrier per K-tile.
//
// MEASURED (MI355X, tools/exp_gemm.sh, the 1024-row head contractions; rocprofv3 device durations): 20.0 us (logits) /
// 21.5 us (value projection) against 20.4 / 20.8 us for the f32 kernel -- no gain.  PMC: matrix pipe 19 % busy, but the
// plane round trip moves 6 bytes per element through LDS on top of the fp32 image (DMA 16 KiB + plane writes 24 KiB +
// plane reads 72 KiB + fragment reads 16 KiB per K-tile): the K loop is LDS-bound at the same ~0.6 us per tile the f32
// kernel spends in its matrix pipe.  32-row tiles (two workgroups per CU) move 40 % more LDS bytes per output and are
// slower (24.7 us).  What would pay is operands that ARRIVE split (planes written by the producing kernels): not done.
#include "ep_side.h"

namespace ep {

typedef __attribute__((address_space(3))) void* x3_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* x3_gptr_t;
typedef __bf16 x3_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf2 __attribute__((ext_vector_type(2)));
typedef unsigned x3_u4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void x3_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void x3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// two fp32 values -> one register holding their bf16 roundings (element 0 in the low half): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned x3_pack_rne(float v0, float v1) {
  typedef float x3_f2 __attribute__((ext_vector_type(2)));
  const x3_f2 v = {v0, v1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_bf2));
}
// two fp32 values -> their three bf16 terms, packed
__device__ __forceinline__ void x3_split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  h = x3_pack_rne(v0, v1);
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);          // exact
  m = x3_pack_rne(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);          // exact, <= 8 bits
  l = x3_pack_rne(s0, s1);
}
__device__ __forceinline__ f4v x3_mfma(x3_u4 a, x3_u4 b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3_bf8, a), __builtin_bit_cast(x3_bf8, b), c, 0, 0, 0);
}

#ifndef X3_ABLATE
#define X3_ABLATE 0                                  // diagnostic builds: 1 operand ring only (no arithmetic), 2 arithmetic only
#endif
// Tile rows BM = 64 (one 112-KiB workgroup per CU) or 32 (72 KiB: two workgroups per CU).  A K-tile step of one
// workgroup is a chain of latencies -- barrier, LDS reads, 6-deep MFMA chains, split, LDS writes, barrier -- that neither
// fills the matrix pipe (19 % busy at BM = 64) nor the LDS (27 %); the head's 1024-row contractions are 256 tiles of 64 x 64,
// i.e. ONE such workgroup per CU.  With 32-row tiles they are 512 workgroups, two per CU, each hiding the other's chain.
template <int BM> struct X3Geo {
  static constexpr int NA = BM / 16;                 // 16-row blocks of the A image (4 or 2); the B image always has 4
  static constexpr int OPA = BM * BK * 4;            // bytes of the fp32 A image
  static constexpr int OPB = 64 * BK * 4;            // bytes of the fp32 B image
  static constexpr int STB = OPA + OPB;              // bytes per fp32 ring stage
  static constexpr int UNITS = NA + 4;               // (operand, block) split units per K-tile
  static constexpr int PLB = UNITS * 3 * 1024;       // bytes per plane buffer: units x 3 terms x 1 KiB
  static constexpr int NST = BM == 64 ? 4 : 3;       // fp32 ring stages
  static constexpr int LDS = NST * STB + 2 * PLB;    // 112 KiB / 72 KiB
};

template <bool A_K, bool B_K, int BM>
__global__ __launch_bounds__(512) void ep_gemm_x3_kernel(GemmParams p) {
  using G = X3Geo<BM>;
  constexpr int NST = G::NST, X3_STB = G::STB, X3_OPA = G::OPA, X3_PLB = G::PLB, NA = G::NA;
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [NST fp32 stages][2 plane buffers]
  char* const planes = lds + NST * X3_STB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
  // multiply role: BM = 64: rows 32 wm .. +31 (two A blocks), columns 16 wn .. +15; BM = 32: A block wm, B block wn
  const int wm = w >> 2, wn = w & 3;
  // split role: unit w -- units 0 .. NA-1 are the A blocks, NA .. NA+3 the B blocks; BM = 32: waves 6, 7 have none
  const bool s_has = w < G::UNITS;
  const int sop = w < NA ? 0 : 1, sblk = w < NA ? w : (w - NA) & 3;
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int nk = (p.K + BK - 1) / BK;

  // ---- DMA: 8 pieces of 1 KiB per operand image; wave w moves piece w of A and piece w of B (the same swizzled images
  // as ep_gemm_dma_kernel: K layout chunk c of row r in slot c ^ ((r >> 1) & 7), T layout chunk c of k-row k in slot
  // c ^ (4 ((k >> 2) & 1)))
  // `rows`: rows of the operand image (BM for A, 64 for B); a T-layout image has rows/4 chunks per k-row
  auto src_off = [&](bool klay, int rows, int64_t ld, int lim, int ext, int r0, int pos, int k0) -> int64_t {
    if (klay) {
      const int r = pos >> 3, q = pos & 7;
      const int kq = q ^ ((r >> 1) & 7);
      int row = r0 + r; row = row < lim ? row : lim - 1;
      int k = k0 + 4 * kq; k = k < p.K ? k : 0;
      return (int64_t)row * ld + k;
    } else {
      const int cpr = rows >> 2;                      // chunks per k-row (16 or 8)
      const int k = pos / cpr, q = pos % cpr;
      const int c = q ^ (4 * ((k >> 2) & 1));
      int kr = k0 + k; kr = kr < p.K ? kr : p.K - 1;
      int col = r0 + 4 * c; col = col < ext ? col : r0;
      return (int64_t)kr * ld + col;
    }
  };
  // BM = 32: the A image has 4 pieces; waves 4-7 re-copy pieces 0-3 (identical bytes) so that every wave issues two DMA
  // instructions per K-tile and the counted vmcnt waits are the same for all
  const int wa = BM == 64 ? w : (w & 3);
  const int pos = w * 64 + lane, posA = wa * 64 + lane;
  const int64_t srcA = src_off(A_K, BM, p.lda, p.M, p.extA, m0, posA, 0);
  const int64_t srcB = src_off(B_K, 64, p.ldb, p.N, p.extB, n0, pos, 0);
  const int64_t kstepA = A_K ? BK : (int64_t)BK * p.lda;
  const int64_t kstepB = B_K ? BK : (int64_t)BK * p.ldb;
  const bool ktail = (p.K % BK) != 0;
  auto issue = [&](int t) {                          // DMA K-tile t (clamped to the last one) into stage t % NST
    const int tt = t < nk ? t : nk - 1;
    char* st = lds + (t % NST) * X3_STB;
    int64_t oa, ob;
    if (ktail && tt == nk - 1) {
      oa = src_off(A_K, BM, p.lda, p.M, p.extA, m0, posA, tt * BK);
      ob = src_off(B_K, 64, p.ldb, p.N, p.extB, n0, pos, tt * BK);
    } else {
      oa = srcA + tt * kstepA; ob = srcB + tt * kstepB;
    }
    __builtin_amdgcn_global_load_lds((x3_gptr_t)(A + oa), (x3_lds_ptr_t)(st + wa * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((x3_gptr_t)(B + ob), (x3_lds_ptr_t)(st + X3_OPA + w * 1024), 16, 0, 0);
  };

  // ---- split role: fp32 fragment of unit (sop, sblk) -> three bf16x8 operands in the plane buffer ----
  // fragment addressing inside an fp32 operand image, as ep_gemm_dma_kernel (row r of the 64-row image):
  //   K layout: chunk 4g + kk of row r at r*128 + 16*((4g + kk) ^ ((r >> 1) & 7))              (one ds_read_b128 per g)
  //   T layout: value (g, j): k = 16g + 4kk + j, column r at (16g + j)*256 + [4kk*256 + 16*((r >> 2) ^ 4(kk & 1)) + 4(r & 3)]
  const bool s_klay = sop == 0 ? A_K : B_K;
  const int s_rowb = (sop == 0 ? BM : 64) * 4;       // bytes per k-row of a T-layout image (256 or 128)
  int sfrag0, sfrag1;
  {
    const int r = sblk * 16 + i16;
    if (s_klay) {
      sfrag0 = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
      sfrag1 = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
    } else {
      sfrag0 = 4 * kk * s_rowb + 16 * ((r >> 2) ^ (4 * (kk & 1))) + 4 * (r & 3);
      sfrag1 = 0;
    }
  }
  const int sdst = (w * 3) * 1024 + lane * 16;       // unit w: [unit][term][lane]
  auto split_load = [&](int t, f4v (&x)[2]) {       // K-tile t: this unit's fp32 fragment from stage t % NST
    const char* src = lds + (t % NST) * X3_STB + sop * X3_OPA;
    if (s_klay) {                                    // wave-uniform
      x[0] = *reinterpret_cast<const f4v*>(src + sfrag0);
      x[1] = *reinterpret_cast<const f4v*>(src + sfrag1);
    } else {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[g][j] = *reinterpret_cast<const float*>(src + sfrag0 + (16 * g + j) * s_rowb);
    }
  };
  auto split_store = [&](int t, f4v (&x)[2]) {      // split the fragment, write the three terms into plane buffer t & 1
    if (ktail && t == nk - 1) {                      // last tile: k >= K contributes nothing
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[g][j] = (t * BK + 16 * g + 4 * kk + j >= p.K) ? 0.f : x[g][j];
    }
    x3_u4 t3[3];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned h, m, l;
        x3_split2(x[g][2 * q], x[g][2 * q + 1], h, m, l);
        t3[0][2 * g + q] = h; t3[1][2 * g + q] = m; t3[2][2 * g + q] = l;
      }
    char* dst = planes + (t & 1) * X3_PLB + sdst;
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) *reinterpret_cast<x3_u4*>(dst + tm * 1024) = t3[tm];
  };

  // ---- multiply role ----
  constexpr int MB = BM / 32;                        // A blocks per wave (2 or 1)
  f4v acc[MB];
#pragma unroll
  for (int bi = 0; bi < MB; ++bi) acc[bi] = f4v{0.f, 0.f, 0.f, 0.f};
  const int moffA = ((MB * wm) * 3) * 1024 + lane * 16;          // A units MB wm .. MB wm + MB - 1
  const int moffB = ((NA + wn) * 3) * 1024 + lane * 16;          // B unit NA + wn
  auto mult_load = [&](int t, x3_u4 (&a3)[MB][3], x3_u4 (&b3)[3]) {
    const char* pb = planes + (t & 1) * X3_PLB;
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) {
      b3[tm] = *reinterpret_cast<const x3_u4*>(pb + moffB + tm * 1024);
#pragma unroll
      for (int bi = 0; bi < MB; ++bi) a3[bi][tm] = *reinterpret_cast<const x3_u4*>(pb + moffA + (bi * 3 + tm) * 1024);
    }
  };
  auto mult = [&](const x3_u4 (&a3)[MB][3], const x3_u4 (&b3)[3]) {
    // smallest terms first: lo x hi, hi x lo, mid x mid, then the 2^-8 pair, then hi x hi
#pragma unroll
    for (int bi = 0; bi < MB; ++bi) acc[bi] = x3_mfma(a3[bi][2], b3[0], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < MB; ++bi) acc[bi] = x3_mfma(a3[bi][0], b3[2], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < MB; ++bi) acc[bi] = x3_mfma(a3[bi][1], b3[1], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < MB; ++bi) acc[bi] = x3_mfma(a3[bi][1], b3[0], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < MB; ++bi) acc[bi] = x3_mfma(a3[bi][0], b3[1], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < MB; ++bi) acc[bi] = x3_mfma(a3[bi][0], b3[0], acc[bi]);
  };

  // ---- pipeline (two DMA instructions per wave per K-tile; tile t lives in stage t % NST) ----
  // prologue: tiles 0 .. NST-1 in flight, tile 0 landed -> split into plane buffer 0
#pragma unroll
  for (int t = 0; t < NST; ++t) issue(t);
  x3_dma_wait<2 * (NST - 1)>();                      // this wave's pieces of tile 0
  x3_barrier();                                      // ... and every other wave's
  if (s_has) {
    f4v x[2];
    split_load(0, x);
    split_store(0, x);
  }
  // step it: [my pieces of tile it+1 landed] barrier [plane buffer it&1 complete, buffer (it+1)&1 free (multiply(it-1)
  // done), fp32 tile it+1 complete, stage it % NST free (split(it) done)] -> refill that stage with tile it+NST, split
  // tile it+1 into the other plane buffer, multiply tile it.  The LDS reads of both roles are issued first; the split
  // arithmetic (vector ALU) and the MFMAs then share one basic block for the scheduler to interleave.
  for (int it = 0; it < nk; ++it) {
    x3_dma_wait<2 * (NST - 2)>();
    x3_barrier();
    x3_u4 a3[MB][3], b3[3];
    f4v x[2];
    mult_load(it, a3, b3);
    const int tn = it + 1 < nk ? it + 1 : it;        // the last step re-splits its own tile into the free buffer: unused
    if (BM == 64 || s_has) split_load(tn, x);
    if (X3_ABLATE != 2) issue(it + NST);
    if (X3_ABLATE != 1) {
      mult(a3, b3);
      if (BM == 64 || s_has) split_store(tn + (it + 1 < nk ? 0 : 1), x);
    }
  }
  x3_dma_wait<0>();                                  // redundant prefetches past the last tile: drain before exit

  const int col = n0 + wn * 16 + i16;
  if (col < p.N) {
    const float bv = p.bias ? p.bias[(int64_t)z * p.sBiasz + col] : 0.f;
#pragma unroll
    for (int bi = 0; bi < MB; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * MB + bi) * 16 + kk * 4 + r;
        if (row < p.M) {
          float* c = C + (int64_t)row * p.ldc + col;
          float v = p.alpha * acc[bi][r] + bv;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
  }
}

template <bool A_K, bool B_K, int BM>
static void x3_launch_one(const GemmParams& p, int batch, hipStream_t st) {
  constexpr int lds = X3Geo<BM>::LDS;
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, batch);
  auto k = ep_gemm_x3_kernel<A_K, B_K, BM>;
  static bool attr_set = false;                      // per instantiation: raise the dynamic-LDS limit once
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
}

// 32-row tiles (two workgroups per CU) unless the 64-row grid already has two rounds of workgroups for every CU
void gemm_launch_x3(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  static int force_bm = -1;
  if (force_bm < 0) { const char* e = getenv("EP_GEMM_X3_BM"); force_bm = e ? atoi(e) : 0; }
  const long tiles64 = (long)((p.N + BN - 1) / BN) * ((p.M + 63) / 64) * batch;
  const bool bm32 = force_bm ? force_bm == 32 : tiles64 < 4L * cu_count();
#define EP_GEMM_LAUNCH(AK, BK_) { if (bm32) x3_launch_one<AK, BK_, 32>(p, batch, st); else x3_launch_one<AK, BK_, 64>(p, batch, st); }
  if (a_k && b_k) EP_GEMM_LAUNCH(true, true)
  else if (a_k && !b_k) EP_GEMM_LAUNCH(true, false)
  else if (!a_k && b_k) EP_GEMM_LAUNCH(false, true)
  else EP_GEMM_LAUNCH(false, false)
#undef EP_GEMM_LAUNCH
}

}  // namespace ep
