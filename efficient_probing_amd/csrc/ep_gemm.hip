// Exact-fp32 tensor contractions of the EP head on the gfx950 matrix cores
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bit-identical to an fmaf chain).
//
// Used for: the per-query value projection y = P Wv_q^T and its two gradients
// (reference poolings/ep.py:40 after the pool-then-project refactoring), and the classifier
// Linear(Dp, C) forward / dW / dz (reference probe_heads.py:76).
//
//   C[z][m][n] (+)= alpha * sum_k A[z](m,k) * B[z](k,n)  (+ bias[n])
//
// Each operand is either contiguous along K ("K" layout: element (r,k) at r*ld + k) or along its
// own free dimension ("T" layout: element (r,k) at k*ld + r).  A 64x64 output tile per
// workgroup (4 waves, 2x2, each 32x32 = 2x2 MFMA blocks), K-step 32, register-prefetch double
// buffering, LDS images padded so that every fragment read (ds_read_b32) is conflict-free:
//   K layout -> LDS [64][32+2]   (lane (i,kk) reads row i, column 4s+kk : bank 2i+kk+4s)
//   T layout -> LDS [32][64+16]  (lane (i,kk) reads row 4s+kk, column i : bank 16kk+i)
#include "ep_side.h"
#include "ep_gemm_dma.h"

#include "ep_wgrad3.h"

namespace ep {

// tile body: ep_side.h (gemm_tile)
template <bool A_K, bool B_K, bool VEC, int BMT>
__global__ __launch_bounds__(256) void ep_gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][LDS_OPERAND];   // [buffer][A|B]
  gemm_tile<A_K, B_K, VEC, BMT>(p, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant (8 waves): waves 4-7 only stage operands (global -> registers, three tiles
// ahead -> LDS, one tile ahead), waves 0-3 only read LDS and issue MFMAs.  The staging work of the
// next tile runs on the same SIMDs in the shadow of the current tile's MFMAs, and the matrix waves
// never wait on global memory: one s_barrier per K-tile is their only synchronisation.
// ---------------------------------------------------------------------------------------------

template <bool A_K, bool B_K, bool VEC>
__global__ __launch_bounds__(512) void ep_gemm_ws_kernel(GemmParams p) {
  constexpr int BMT = 64, MI = 2;
  __shared__ __attribute__((aligned(16))) float lds[2][2][LDS_OPERAND];
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tid = threadIdx.x & 255;                 // index inside the role group
  const int lane = tid & 63;
  const int m0 = blockIdx.y * BMT, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int nk = (p.K + BK - 1) / BK;
  auto tile_k0 = [&](int it) { return (it < nk ? it : nk - 1) * BK; };

  if (w >= 4) {
    // ------------------------------ loader waves ------------------------------
    f4v ra[3][2], rb[3][2];
    auto gload = [&](int k0, f4v (&xa)[2], f4v (&xb)[2]) {
      if (A_K) load_K<VEC, BMT>(A, p.lda, p.M, p.K, m0, k0, tid, xa);
      else load_T<VEC, BMT>(A, p.lda, p.extA, p.K, m0, k0, tid, xa);
      if (B_K) load_K<VEC, BN>(B, p.ldb, p.N, p.K, n0, k0, tid, xb);
      else load_T<VEC, BN>(B, p.ldb, p.extB, p.K, n0, k0, tid, xb);
    };
    auto lstore = [&](int buf, const f4v (&xa)[2], const f4v (&xb)[2], int k0) {
      if (A_K) store_K<BMT>(lds[buf][0], tid, xa, p.M, p.K, m0, k0); else store_T<BMT>(lds[buf][0], tid, xa, p.extA, p.K, m0, k0);
      if (B_K) store_K<BN>(lds[buf][1], tid, xb, p.N, p.K, n0, k0); else store_T<BN>(lds[buf][1], tid, xb, p.extB, p.K, n0, k0);
    };
    gload(0, ra[0], rb[0]);
    gload(tile_k0(1), ra[1], rb[1]);
    gload(tile_k0(2), ra[2], rb[2]);
    lstore(0, ra[0], rb[0], 0);
    ws_barrier();                                    // tile 0 is staged
#define EP_WS_LOAD_STEP(IT, J)                                                 \
    {                                                                          \
      gload(tile_k0((IT) + 3), ra[J], rb[J]);                                  \
      lstore(((IT) + 1) & 1, ra[((J) + 1) % 3], rb[((J) + 1) % 3], ((IT) + 1) * BK); \
      ws_barrier();                                  /* tile IT+1 staged; matrix waves done with tile IT */ \
    }
    int it = 0;
    for (; it + 2 < nk; it += 3) {
      EP_WS_LOAD_STEP(it, 0)
      EP_WS_LOAD_STEP(it + 1, 1)
      EP_WS_LOAD_STEP(it + 2, 2)
    }
    if (it < nk) {
      EP_WS_LOAD_STEP(it, 0)
      if (it + 1 < nk) EP_WS_LOAD_STEP(it + 1, 1)
    }
#undef EP_WS_LOAD_STEP
    return;
  }
  // ------------------------------ matrix waves ------------------------------
  const int wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, kk = lane >> 4;
  f4v acc[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  ws_barrier();
  for (int it = 0; it < nk; ++it) {
    const float* As = lds[it & 1][0];
    const float* Bs = lds[it & 1][1];
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      float af[MI], bf[2];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * 32 + mi * 16 + i16;
        af[mi] = A_K ? As[row * LDK + 4 * s + kk] : As[(4 * s + kk) * LDT + row];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = wn * 32 + ni * 16 + i16;
        bf[ni] = B_K ? Bs[col * LDK + 4 * s + kk] : Bs[(4 * s + kk) * LDT + col];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    ws_barrier();
  }
  {
    f4v blk[MI * 2]; int rb[MI * 2], cb[MI * 2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + wm * 32 + mi * 16; cb[mi * 2 + ni] = n0 + wn * 32 + ni * 16;
      }
    store_acc_blocks<MI * 2>(p, C, z, rb, cb, blk, kk, i16);
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (the default for 16-byte aligned operands).  The stand-alone contractions of the head run one
// 64x64 tile per CU, so nothing hides a workgroup's own latencies: with register staging the K loop ran at ~2x its
// MFMA time, waiting on global loads two K-tiles deep.  Here the operand tiles go from L2 straight into a ring of
// NST LDS stages (global_load_lds_dwordx4, 1 KiB per wave-instruction, NST-1 K-tiles in flight, exact counted
// vmcnt waits), no staging registers.  WS = wave-specialised (8 waves): waves 4-7 only issue the DMA (an LDS-DMA
// instruction costs its issuing wave ~100 cycles, which a matrix wave would spend not issuing MFMAs), waves 0-3
// only read fragments and multiply; !WS = 4 symmetric waves doing both.
//   * LDS images are unpadded and XOR-swizzled -- the swizzle is applied for free on the DMA source address:
//       K layout (64 rows x 8 chunks of 16 B): chunk c of row r sits in slot c ^ ((r >> 1) & 7); a lane reads its
//         four k-values with ONE ds_read_b128, conflict-free for the 4 x 16 lane groups of that instruction;
//       T layout (32 k-rows x 16 chunks): chunk c of k-row k sits in slot c ^ (4 * ((k >> 2) & 1)); ds_read_b32,
//         conflict-free for its 2 x 32 lane groups.
//   * k-mapping inside a K-tile: MFMA step (g, j) (g = 0..1, j = 0..3) feeds lane group kk with k = 16 g + 4 kk + j,
//     the same for both operands (any bijection is valid for the contraction).
//   * fragments of tile it+1 are read into a second register set while tile it is multiplied (a whole K-tile of
//     MFMAs hides the LDS latency); one s_barrier per K-tile.
//   * K tail (K % 32 != 0): sources are clamped to valid addresses and the out-of-range k fragments are zeroed in
//     registers on the last tile.  Rows / columns beyond M / N are clamped on load and masked at the store.
// ---------------------------------------------------------------------------------------------
template <bool A_K, bool B_K, int NST, bool WS>
__global__ __launch_bounds__(WS ? 512 : 256) void ep_gemm_dma_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * 2 * 64 * BK * 4];
  gemm_dma_tile<A_K, B_K, NST, WS>(p, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// ---------------------------------------------------------------------------------------------
// 32 x 96 tiles for K/K-layout contractions whose N is a multiple of 96 but not of 64 -- the per-query value projection
// y_q = P_q Wv_q^T (N = Dp / Q = 96 at 256x768, Q = 8): 64x64 tiles leave every second tile half empty (a quarter of the
// matrix time).  Same ring, swizzle, k-mapping and wave specialisation as ep_gemm_dma_kernel<true, true, 4, true>; a stage
// holds a 32-row A image (4 KiB) and a 96-row B image (12 KiB) = the same 16 one-KiB DMA pieces, four per loader wave;
// matrix wave (wm, wn) owns 16 rows x 48 columns = 1 x 3 MFMA blocks.  K % 32 == 0, N % 96 == 0 (host-checked).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void ep_gemm_kk96_kernel(GemmParams p) {
  constexpr int NST = 4;
  constexpr int STB = 16 * 1024;                     // bytes per ring stage: A image 4 KiB | B image 12 KiB
  constexpr int AIMG = 4 * 1024;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wall >= 4;
  const int w = wall & 3;
  const int wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 96;
  const int z = blockIdx.z;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int nk = p.K / BK;

  if (loader) {
    // piece pc = w + 4 jj of the stage: positions pc*64 + lane -> image row r = pos >> 3 (A rows 0..31, then B rows 0..95),
    // slot q = pos & 7 holding k-chunk q ^ ((r >> 1) & 7) of that row
    const float* src[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int pos = (w + 4 * jj) * 64 + lane;
      int r = pos >> 3;
      const int q = pos & 7;
      const bool isA = r < 32;
      if (!isA) r -= 32;
      const int kq = q ^ ((r >> 1) & 7);
      if (isA) { int row = m0 + r; row = row < p.M ? row : p.M - 1; src[jj] = A + (int64_t)row * p.lda + 4 * kq; }
      else src[jj] = B + (int64_t)(n0 + r) * p.ldb + 4 * kq;
    }
    auto issue = [&](int t) {
      const int tt = t < nk ? t : nk - 1;            // past the end: the last tile again (uniform vmcnt arithmetic)
      char* st = lds + (t % NST) * STB;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[jj] + tt * BK), (lds_ptr_t)(st + (w + 4 * jj) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t);
    dma_wait<4 * (NST - 2)>();
    ws_barrier();                                    // tile 0 landed
    for (int it = 0; it < nk; ++it) {
      dma_wait<4 * (NST - 3)>();                     // tile it+1 landed (this wave's pieces)
      ws_barrier();                                  // ... everyone's; the matrix waves hold tile `it` in registers
      issue(it + NST - 1);
    }
    dma_wait<0>();
    return;
  }

  int fragA[2], fragB[3][2];
  {
    const int r = wm * 16 + i16;
    fragA[0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
    fragA[1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int bi = 0; bi < 3; ++bi) {
    const int r = wn * 48 + bi * 16 + i16;
    fragB[bi][0] = AIMG + r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
    fragB[bi][1] = AIMG + r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
  }
  f4v fa[2][2], fb[2][3][2];                         // [set][g], [set][block][g]
  auto read_frags = [&](int stage, f4v (&xa)[2], f4v (&xb)[3][2]) {
    const char* sb = lds + stage * STB;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      xa[g] = *reinterpret_cast<const f4v*>(sb + fragA[g]);
#pragma unroll
      for (int bi = 0; bi < 3; ++bi) xb[bi][g] = *reinterpret_cast<const f4v*>(sb + fragB[bi][g]);
    }
  };
  f4v acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) acc[b] = f4v{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const f4v (&xa)[2], const f4v (&xb)[3][2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int bi = 0; bi < 3; ++bi)
          acc[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g][j], xb[bi][g][j], acc[bi], 0, 0, 0);
  };
  ws_barrier();                                      // tile 0 landed
  read_frags(0, fa[0], fb[0]);
#define EP_KK96_STEP(IT, F)                                      \
  {                                                              \
    __builtin_amdgcn_s_waitcnt(0xc07f);                          \
    ws_barrier();                                                \
    read_frags(((IT) + 1) % NST, fa[(F) ^ 1], fb[(F) ^ 1]);      \
    __builtin_amdgcn_sched_barrier(0);                           \
    multiply(fa[F], fb[F]);                                      \
    __builtin_amdgcn_sched_barrier(0);                           \
  }
  int it = 0;
  for (; it + 1 < nk; it += 2) {
    EP_KK96_STEP(it, 0)
    EP_KK96_STEP(it + 1, 1)
  }
  if (it < nk) EP_KK96_STEP(it, 0)
#undef EP_KK96_STEP
  int rb[3], cb[3];
#pragma unroll
  for (int bi = 0; bi < 3; ++bi) { rb[bi] = m0 + wm * 16; cb[bi] = n0 + wn * 48 + bi * 16; }
  store_acc_blocks<3>(p, C, z, rb, cb, acc, kk, i16);
}

// The same 32 x 96 tiling with the B operand in T layout (k-rows of N contiguous columns): dz = dlogits Wc at
// 1024 x 768 is 16 x 12 = 192 tiles of 64 x 64 on 256 CUs, and 32 x 8 = 256 tiles of 32 x 96.  B image: 32 k-rows of 96
// floats (24 chunks of 16 B, chunk c of k-row k in slot c ^ 4((k >> 2) & 1): the T-layout swizzle of the 64-column
// kernel; rows are 96 words apart, k-rows 4 apart still alias modulo 64 banks, so the same two-group argument holds).
// K tail (K % 32 != 0; K % 4 == 0): sources clamped, out-of-range k zeroed in registers on the last tile.
__global__ __launch_bounds__(512) void ep_gemm_kt96_kernel(GemmParams p) {
  constexpr int NST = 4;
  constexpr int STB = 16 * 1024;
  constexpr int AIMG = 4 * 1024;
  constexpr int ROWB = 96 * 4;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wall >= 4;
  const int w = wall & 3;
  const int wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 96;
  const int z = blockIdx.z;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int nk = (p.K + BK - 1) / BK;
  const bool ktail = (p.K % BK) != 0;

  if (loader) {
    auto src_of = [&](int jj, int k0) -> const float* {
      const int pos = (w + 4 * jj) * 64 + lane;
      if (pos < 256) {                                 // A image, K layout: row r, slot q holds k-chunk q ^ ((r >> 1) & 7)
        const int r = pos >> 3, q = pos & 7;
        const int kq = q ^ ((r >> 1) & 7);
        int row = m0 + r; row = row < p.M ? row : p.M - 1;
        int k = k0 + 4 * kq; k = k < p.K ? k : 0;
        return A + (int64_t)row * p.lda + k;
      }
      const int pb = pos - 256;                        // B image, T layout: k-row k, slot q holds column chunk q ^ 4((k >> 2) & 1)
      const int k = pb / 24, q = pb - 24 * k;
      const int c = q ^ (4 * ((k >> 2) & 1));
      int kr = k0 + k; kr = kr < p.K ? kr : p.K - 1;
      return B + (int64_t)kr * p.ldb + n0 + 4 * c;
    };
    const float* src[4];
    int64_t step[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      src[jj] = src_of(jj, 0);
      step[jj] = ((w + 4 * jj) * 64 + lane) < 256 ? (int64_t)BK : (int64_t)BK * p.ldb;
    }
    auto issue = [&](int t) {
      const int tt = t < nk ? t : nk - 1;
      char* st = lds + (t % NST) * STB;
      if (ktail && tt == nk - 1) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          __builtin_amdgcn_global_load_lds((gptr_t)src_of(jj, tt * BK), (lds_ptr_t)(st + (w + 4 * jj) * 1024), 16, 0, 0);
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          __builtin_amdgcn_global_load_lds((gptr_t)(src[jj] + tt * step[jj]), (lds_ptr_t)(st + (w + 4 * jj) * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t);
    dma_wait<4 * (NST - 2)>();
    ws_barrier();
    for (int it = 0; it < nk; ++it) {
      dma_wait<4 * (NST - 3)>();
      ws_barrier();
      issue(it + NST - 1);
    }
    dma_wait<0>();
    return;
  }

  int fragA[2], fragB[3];
  {
    const int r = wm * 16 + i16;
    fragA[0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
    fragA[1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int bi = 0; bi < 3; ++bi) {
    const int r = wn * 48 + bi * 16 + i16;             // column inside the tile
    fragB[bi] = AIMG + 4 * kk * ROWB + 16 * ((r >> 2) ^ (4 * (kk & 1))) + 4 * (r & 3);
  }
  f4v fa[2][2], fb[2][3][2];
  auto read_frags = [&](int stage, f4v (&xa)[2], f4v (&xb)[3][2]) {
    const char* sb = lds + stage * STB;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      xa[g] = *reinterpret_cast<const f4v*>(sb + fragA[g]);
#pragma unroll
      for (int bi = 0; bi < 3; ++bi)
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[bi][g][j] = *reinterpret_cast<const float*>(sb + fragB[bi] + (16 * g + j) * ROWB);
    }
  };
  auto zero_tail = [&](int k0, f4v (&xa)[2], f4v (&xb)[3][2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool out = k0 + 16 * g + 4 * kk + j >= p.K;
        xa[g][j] = out ? 0.f : xa[g][j];
#pragma unroll
        for (int bi = 0; bi < 3; ++bi) xb[bi][g][j] = out ? 0.f : xb[bi][g][j];
      }
  };
  f4v acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) acc[b] = f4v{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const f4v (&xa)[2], const f4v (&xb)[3][2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int bi = 0; bi < 3; ++bi)
          acc[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g][j], xb[bi][g][j], acc[bi], 0, 0, 0);
  };
  ws_barrier();
  read_frags(0, fa[0], fb[0]);
  if (ktail && nk == 1) { __builtin_amdgcn_s_waitcnt(0xc07f); zero_tail(0, fa[0], fb[0]); }
#define EP_KT96_STEP(IT, F)                                      \
  {                                                              \
    __builtin_amdgcn_s_waitcnt(0xc07f);                          \
    ws_barrier();                                                \
    read_frags(((IT) + 1) % NST, fa[(F) ^ 1], fb[(F) ^ 1]);      \
    __builtin_amdgcn_sched_barrier(0);                           \
    multiply(fa[F], fb[F]);                                      \
    __builtin_amdgcn_sched_barrier(0);                           \
    if (ktail && (IT) + 1 == nk - 1) { __builtin_amdgcn_s_waitcnt(0xc07f); zero_tail(((IT) + 1) * BK, fa[(F) ^ 1], fb[(F) ^ 1]); } \
  }
  int it = 0;
  for (; it + 1 < nk; it += 2) {
    EP_KT96_STEP(it, 0)
    EP_KT96_STEP(it + 1, 1)
  }
  if (it < nk) EP_KT96_STEP(it, 0)
#undef EP_KT96_STEP
  int rb[3], cb[3];
#pragma unroll
  for (int bi = 0; bi < 3; ++bi) { rb[bi] = m0 + wm * 16; cb[bi] = n0 + wn * 48 + bi * 16; }
  store_acc_blocks<3>(p, C, z, rb, cb, acc, kk, i16);
  if (p.cs_out) {
    // column statistics of this 32-row tile (GemmParams.cs_out): a lane holds rows 4 kk + r of a 16-row block in one
    // column; sum its rows, fold the four lane groups, then the two row halves (wm) through LDS -- fixed order
    float s1[3], s2[3];
#pragma unroll
    for (int bi = 0; bi < 3; ++bi) {
      const int col = n0 + wn * 48 + bi * 16 + i16;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 16 + 4 * kk + r;
        const float zz = p.cs_z[(int64_t)(row < p.M ? row : p.M - 1) * p.ldc + col];
        const float v = row < p.M ? p.alpha * acc[bi][r] : 0.f;
        a += v; b = fmaf(v, zz, b);
      }
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      s1[bi] = a; s2[bi] = b;
    }
    float* ex = reinterpret_cast<float*>(lds);          // [wm][stat][96]; the ring is idle (the loader waves drained it)
    ws_barrier();
    if (kk == 0) {
#pragma unroll
      for (int bi = 0; bi < 3; ++bi) {
        ex[(wm * 2 + 0) * 96 + wn * 48 + bi * 16 + i16] = s1[bi];
        ex[(wm * 2 + 1) * 96 + wn * 48 + bi * 16 + i16] = s2[bi];
      }
    }
    ws_barrier();
    if (wm == 0 && kk == 0) {
#pragma unroll
      for (int bi = 0; bi < 3; ++bi) {
        const int c = wn * 48 + bi * 16 + i16;
        float* o = p.cs_out + ((int64_t)blockIdx.y * 2) * p.N + n0 + c;
        o[0] = ex[0 * 96 + c] + ex[2 * 96 + c];
        o[p.N] = ex[1 * 96 + c] + ex[3 * 96 + c];
      }
    }
  }
}

// K/T layout on 32 x 96 tiles: where the 64 x 64 grid is a partial round of the chip and the 32 x 96 grid a whole one
static bool gemm_kt96_ok(bool a_k, bool b_k, const GemmParams& p, int batch) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_GEMM_KT96"); on = e ? atoi(e) : 1; }
  if (!on || !a_k || b_k || p.N % 96 != 0 || p.K < BK || p.K % 4 != 0 || p.M < 1 || p.npers > 1) return false;
  const long cus = cu_count();
  const long t64 = (long)((p.N + 63) / 64) * ((p.M + 63) / 64) * batch;
  const long t96 = (long)(p.N / 96) * ((p.M + 31) / 32) * batch;
  // one partial round of 64 x 64 tiles against whole rounds of the smaller ones (same matrix work per round)
  return t64 < cus && t96 % cus == 0 && (p.K + BK - 1) / BK >= 8;
}

bool gemm_colstats_ok(bool a_k, bool b_k, const GemmParams& p, int batch) {
  static int use_dma = -1;
  if (use_dma < 0) { const char* e = getenv("EP_GEMM_DMA"); use_dma = e ? atoi(e) : 1; }
  static int force_bm = -1;
  if (force_bm < 0) { const char* e = getenv("EP_GEMM_BM"); force_bm = e ? atoi(e) : 0; }
  const bool vec = aligned16(p.A) && aligned16(p.B) && p.lda % 4 == 0 && p.ldb % 4 == 0 && p.K % 4 == 0 && p.extB % 4 == 0;
  return batch == 1 && use_dma && !force_bm && vec && !p.skws && p.M % 32 == 0 && gemm_kt96_ok(a_k, b_k, p, batch);
}

static bool gemm_kk96_ok(bool a_k, bool b_k, const GemmParams& p, int batch) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_GEMM_KK96"); on = e ? atoi(e) : 1; }
  if (!on || !a_k || !b_k || p.N % 96 != 0 || p.N % 64 == 0 || p.K % BK != 0 || p.K < BK || p.M < 1) return false;
  // worth it where the 64x64 tiling wastes tiles and the 32-row grid still fills the chip
  const long tiles = (long)(p.N / 96) * ((p.M + 31) / 32) * batch;
  // from a quarter of the CU count on (EP_GEMM_KK96_MIN_PCT; until round 5: all of it): the value projection of 512 / 256 images
  // (128 / 64 tiles of 32 x 96) -- the per-GPU batches of the 8- and 16-GPU protocol points -- took 18.8 us on the 64-column
  // kernel, a third of whose columns are padding there: 256 x 768 steps 0.2538 -> 0.2497 ms (B = 512), 0.1969 -> 0.1926 (B = 256)
  static int min_pct = -1;
  if (min_pct < 0) { const char* e = getenv("EP_GEMM_KK96_MIN_PCT"); min_pct = e ? atoi(e) : 25; }
  return tiles * 100 >= (long)cu_count() * min_pct;
}

static void gemm_launch_dma(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  const int npers = p.npers > 1 ? p.npers : 1;
  dim3 grid(((p.N + BN - 1) / BN + npers - 1) / npers, (p.M + 63) / 64, batch);
  // 4 ring stages (64 KiB: two workgroups per CU on large grids), wave-specialised: measured fastest of
  // {3, 4, 5 stages} x {specialised, symmetric} on both the 256-tile head contractions and the 65536-row AbMILP ones
#define EP_GEMM_LAUNCH(AK, BK_) hipLaunchKernelGGL((ep_gemm_dma_kernel<AK, BK_, 4, true>), grid, dim3(512), 0, st, p)
  if (a_k && b_k) EP_GEMM_LAUNCH(true, true);
  else if (a_k && !b_k) EP_GEMM_LAUNCH(true, false);
  else if (!a_k && b_k) EP_GEMM_LAUNCH(false, true);
  else EP_GEMM_LAUNCH(false, false);
#undef EP_GEMM_LAUNCH
}

static void gemm_launch_ws(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + 63) / 64, batch);
#define EP_GEMM_LAUNCH(AK, BK_) hipLaunchKernelGGL((ep_gemm_ws_kernel<AK, BK_, true>), grid, dim3(512), 0, st, p)
  if (a_k && b_k) EP_GEMM_LAUNCH(true, true);
  else if (a_k && !b_k) EP_GEMM_LAUNCH(true, false);
  else if (!a_k && b_k) EP_GEMM_LAUNCH(false, true);
  else EP_GEMM_LAUNCH(false, false);
#undef EP_GEMM_LAUNCH
}

template <int BMT>
static void gemm_launch(bool a_k, bool b_k, bool vec, const GemmParams& p, int batch, hipStream_t st) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + BMT - 1) / BMT, batch);
#define EP_GEMM_LAUNCH(AK, BK_, V) hipLaunchKernelGGL((ep_gemm_kernel<AK, BK_, V, BMT>), grid, dim3(256), 0, st, p)
  if (vec) {
    if (a_k && b_k) EP_GEMM_LAUNCH(true, true, true);
    else if (a_k && !b_k) EP_GEMM_LAUNCH(true, false, true);
    else if (!a_k && b_k) EP_GEMM_LAUNCH(false, true, true);
    else EP_GEMM_LAUNCH(false, false, true);
  } else {
    if (a_k && b_k) EP_GEMM_LAUNCH(true, true, false);
    else if (a_k && !b_k) EP_GEMM_LAUNCH(true, false, false);
    else if (!a_k && b_k) EP_GEMM_LAUNCH(false, true, false);
    else EP_GEMM_LAUNCH(false, false, false);
  }
#undef EP_GEMM_LAUNCH
}

// vector loads need 16-byte aligned rows on every operand
static bool vec_ok(const float* ptr, int64_t ld, int64_t sz, int inner) {
  return aligned16(ptr) && ld % 4 == 0 && sz % 4 == 0 && inner % 4 == 0;
}

// can this contraction run as a side task of the second token pass (ep_side.h: T/T layout, vector loads)?
bool gemm_side_ok(const GemmParams& p, bool a_k, bool b_k) {
  return !a_k && !b_k && p.M > 0 && p.N > 0 && p.K > 0 && !p.bias &&
         vec_ok(p.A, p.lda, p.sAz, p.extA) && vec_ok(p.B, p.ldb, p.sBz, p.extB);
}

// Few output tiles and a very long K (weight gradients over all B N token rows): the tiles cannot fill the chip, so K is cut
// into `splits` slices that run as the batch dimension into the caller's scratch and are summed in slice order
// (ep_reduce_partials_kernel: deterministic).  Needs a contiguous C, no bias, one batch.
static bool b3_wide_ok(const GemmParams& p, bool a_k, bool b_k, int batch);
// out = (accumulate ? out : 0) + sum of the K slices in slice order (+ bias[col]): the slice sum of an ACTIVATION x WEIGHT contraction
// (gemm_split_k, the long-K rule), whose bias and residual the slices cannot carry themselves.  n4 / ncol4: float4 counts.
__global__ __launch_bounds__(256) void ep_splitk_sum_bias_kernel(const float* __restrict__ part, int nparts, int64_t n4,
                                                               const float* __restrict__ bias, int ncol4, int accumulate,
                                                               float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f4* p4 = reinterpret_cast<const f4*>(part);
  f4 s = p4[i];
  for (int k = 1; k < nparts; ++k) s += p4[(int64_t)k * n4 + i];
  if (bias) s += reinterpret_cast<const f4*>(bias)[i % ncol4];
  if (accumulate) s += reinterpret_cast<const f4*>(out)[i];
  reinterpret_cast<f4*>(out)[i] = s;
}
static int gemm_split_k(bool a_k, bool b_k, const GemmParams& p, hipStream_t st, bool* done) {
  *done = false;
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_GEMM_SPLITK"); on = e ? atoi(e) : 1; }
  // LONG-K activation x weight contractions (round 6, r6.15): the MLP layers of the SigLIP / V-JEPA / CaiT heads contract 1024 rows over
  // the hidden width 3072 into 768 columns -- 192 tiles of 64 x 64, each a 96-K-tile latency chain: 60 us forward (fc2) and 50 - 100 us
  // backward (dpre W1) for 4.8 GFLOP.  With a slice scratch from the caller they run as K slices on the bf16 x3 tile (three
  // workgroups per CU), summed in slice order together with bias and residual.  EP_SPLITK_LONG=0: off.
  static int long_rule = -1;
  if (long_rule < 0) { const char* e = getenv("EP_SPLITK_LONG"); long_rule = e ? atoi(e) : 1; }
  const bool lng = long_rule && a_k && p.N % 4 == 0 && (!p.bias || (aligned16(p.bias) && p.sBiasz == 0));
  if (!on || !p.skws || (p.bias && !lng) || p.ldc != p.N || p.alpha != 1.0f || p.K < 2048 || ((int64_t)p.M * p.N) % 4 != 0) return 0;
  const bool wide = b3_wide_ok(p, a_k, b_k, 1);        // 128 x 128 tiles (ep_wgrad3.h: gemm_tile_b3w): a quarter of the tiles
  const long tiles = wide ? (long)((p.N + 127) / 128) * ((p.M + 127) / 128) : (long)((p.N + BN - 1) / BN) * ((p.M + 63) / 64);
  const long cus = cu_count();
  if (tiles >= 2 * cus) return 0;
  static int few_model = -1;                           // EP_SPLITK_FEW=0: the CU model and the K >= 8192 floor for every tile count
  if (few_model < 0) { const char* e = getenv("EP_SPLITK_FEW"); few_model = e ? atoi(e) : 1; }
  const bool few = few_model && !wide && !a_k && !b_k && tiles * 4 <= cus;
  const bool longk = lng && tiles < cus && gemm_b3_on();
  if (a_k && !longk) return 0;
  if (p.K < 8192 && !few && !longk) return 0;
  const int min_slice = (few || longk) ? 512 : 1024;   // rows per slice
  // the matrix pipe of a CU is shared by its workgroups: makespan ~ ceil(tiles * s / CUs) * (K / s); pick the s <= 16 with
  // the smallest one (ties: fewer slices) among those that divide K into whole K-tiles and fit the scratch
  const int64_t mn = (int64_t)p.M * p.N;
  int splits = 1;
  // (VERY few tiles -- at most a quarter of the CUs, e.g. the 256 x 768 position-embedding gradient of the CLIP head over its
  // 4096 B H rows: 48 tiles, each a 128-K-tile latency chain, 130 - 165 us for 1.6 GFLOP on the side queue with the optimizer
  // waiting for it.  The bf16 x3 tile is latency-bound there and three workgroups share a CU without slowing each other, so the
  // slices are counted against 3 CUs' worth of slots, from K = 2048, down to 512 rows per slice.)
  const long slots = (few || longk) ? 3 * cus : cus;
  double best = (double)((tiles + slots - 1) / slots);
  for (int sp = 2; sp <= 16; ++sp) {
    if ((size_t)sp * mn > p.skws_floats || p.K % (sp * BK) != 0 || p.K / sp < min_slice) continue;
    const double cost = (double)((tiles * sp + slots - 1) / slots) / sp;
    if (cost < best * 0.95) { best = cost; splits = sp; }
  }
  static int one_model = -1;                       // EP_SPLITK_WIDE1=0: the two-per-CU rule below for the single-product tile too
  if (one_model < 0) { const char* e = getenv("EP_SPLITK_WIDE1"); one_model = e ? atoi(e) : 1; }
  if (wide && one_model && (gemm_arith() == 1 || p.nterms == 1)) {
    // the single-product wide tile (gemm_tile_b3w1) is latency-bound, THREE workgroups per CU: makespan ~ ceil(tiles * s / (3 CUs)) * (K / s).
    // dWqkv of the AbMILP step (243 tiles): 4 slices = 972 workgroups = 1.27 rounds of 768 under the rule below; 8 slices = 2.53
    // rounds of half the length.
    const long slots = 3 * cus;
    double bc = (double)((tiles + slots - 1) / slots);
    int bs = 1;
    for (int sp = 2; sp <= 16; ++sp) {
      if ((size_t)sp * mn > p.skws_floats || p.K % (sp * BK) != 0 || p.K / sp < 1024) continue;
      const double cost = (double)((tiles * sp + slots - 1) / slots) / sp;
      if (cost < bc * 0.95) { bc = cost; bs = sp; }
    }
    splits = bs;
  } else
  if (wide && tiles * splits < 2 * cus) {
    // the wide tile is one 4-wave workgroup with two barriers per K-tile: a CU needs TWO of them to keep its matrix pipe fed
    // (the makespan model above counts whole rounds only) -- the smallest slice count that gives every CU two
    for (int sp = splits + 1; sp <= 16; ++sp) {
      if ((size_t)sp * mn > p.skws_floats || p.K % (sp * BK) != 0 || p.K / sp < 1024) continue;
      splits = sp;
      if (tiles * sp >= 2 * cus) break;
    }
  }
  if (splits < 2) return 0;
  const int kc = p.K / splits;
  GemmParams q = p;
  q.K = kc; q.C = p.skws; q.ldc = p.N; q.sCz = mn; q.accumulate = 0; q.skws = nullptr;
  q.sAz = a_k ? kc : (int64_t)kc * p.lda;
  q.sBz = b_k ? kc : (int64_t)kc * p.ldb;
  if (longk) q.bias = nullptr;
  EP_TRY(gemm(a_k, b_k, q, splits, st));
  if (longk) {
    const int64_t n4 = mn / 4;
    hipLaunchKernelGGL(ep_splitk_sum_bias_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p.skws, splits, n4, p.bias, p.N / 4,
                       p.accumulate, p.C);
    EP_LAUNCH_CHECK("ep_splitk_sum_bias_kernel");
  } else {
    EP_TRY(reduce_partials(p.skws, splits, (int)mn, 1.0f, p.accumulate, p.C, nullptr, st));
  }
  *done = true;
  return 0;
}

// ---- weight-gradient contractions (both operands summed over their SLOW index: T / T layout) on the bf16 matrix cores at
// fp32 accuracy (ep_wgrad3.h): the stand-alone launch of the tile the second token pass runs as side work ----
// EP_B3_STAGES=2 (A/B builds): the two-stage form of the tile (ep_wgrad3.h: gemm_tile_b3g2 -- 60 KiB of LDS, one barrier per
// K-tile, rows fetched two tiles ahead).  Measured SLOWER in round 4 (same box, alternating): AbMILP 9.35 - 9.40 against
// 8.86 - 8.93 ms per step, DINOv2 block 21.7 against 20.3 ms, 196 x 4096 2.38 - 2.41 against 2.365 - 2.374 ms -- two workgroups
// per CU instead of four to five cover less of each other's barriers and LDS latency than the second barrier costs.
#ifndef EP_B3_STAGES
#define EP_B3_STAGES 1
#endif
// Launched 1-D; tile V of the (batch entry, M-tile, N-tile) numbering, N fastest.  xcd (b3_launch): V = (L % 8) * (grid / 8) + L / 8,
// so that XCD L % 8 works through WHOLE batch entries (the images of a batched attention product, the K slices of a weight
// gradient) and the tiles that share an entry's operands share one L2; otherwise V = L, the launch order.
template <bool A_K, bool B_K, int BMT>
__global__ __launch_bounds__(256) void ep_gemm_b3_kernel(GemmParams p, int gx, int gy, unsigned ntiles, int xcd) {
  extern __shared__ __attribute__((aligned(1024))) char lds_b3[];
  const unsigned L = blockIdx.x;
  const unsigned V = xcd ? (L % 8u) * (gridDim.x / 8u) + L / 8u : L;
  if (V >= ntiles) return;
  const int bx = __builtin_amdgcn_readfirstlane((int)(V % (unsigned)gx)), by = __builtin_amdgcn_readfirstlane((int)((V / (unsigned)gx) % (unsigned)gy)),
            bz = __builtin_amdgcn_readfirstlane((int)(V / (unsigned)(gx * gy)));
  if constexpr (EP_B3_STAGES == 2) gemm_tile_b3g2<A_K, B_K, BMT>(p, bx, by, bz, lds_b3);
  else gemm_tile_b3g<A_K, B_K, BMT>(p, bx, by, bz, lds_b3);
}
// LONG weight gradients (T / T, K >= 4096) of at least 1024 x 1024 outputs: 128 x 128 tiles (ep_wgrad3.h: gemm_tile_b3w).
// Measured (ms per step, 256 images, 64 x 64 tiles -> wide): AbMILP 256 x 1152 18.22 -> 16.97 (AMP-bf16 mode 10.59 -> 9.79); at
// D = 768 -- 36 wide tiles per gradient, one or two workgroups per CU after the K split -- it LOSES: DINOv2 block 20.3 -> 21.4,
// DOLG 1.91 -> 1.98, AbMILP 8.75 -> 8.59: hence the size floor.  EP_GEMM_B3_WIDE=0: off; =2: from 256 x 256 (the first form).
// Tile order (round 6): 1-D launch, workgroup L runs on XCD L % 8 and takes tile V = (L % 8) * (grid / 8) + L / 8 of the
// (K slice, M-tile, N-tile) numbering, N fastest: an XCD works through WHOLE K slices, so the 9 x 9 ... 27 x 9 tiles that share
// a slice's operand panels share one L2.  In launch order (tiles of a slice dealt round-robin over the eight L2s) every XCD
// held ~10 tiles of each of ~6 slices and the panels -- 2 x 9 x 302 MB per 1152 x 1152 gradient -- came from HBM.
// EP_B3_WIDE_XCD=0: launch order.
template <bool A_K, bool B_K>
__global__ __launch_bounds__(256) void ep_gemm_b3_wide_kernel(GemmParams p, int mtn, int ntn, unsigned ntiles, int xcd) {
  extern __shared__ __attribute__((aligned(1024))) char lds_b3[];
  const unsigned L = blockIdx.x;
  const unsigned V = xcd ? (L % 8u) * (gridDim.x / 8u) + L / 8u : L;
  if (V >= ntiles) return;
  const int bx = __builtin_amdgcn_readfirstlane((int)(V % (unsigned)ntn)), by = __builtin_amdgcn_readfirstlane((int)((V / (unsigned)ntn) % (unsigned)mtn)),
            bz = __builtin_amdgcn_readfirstlane((int)(V / (unsigned)(mtn * ntn)));
  gemm_tile_b3w<A_K, B_K>(p, bx, by, bz, lds_b3);
}
// ... and its single-product form (AMP-bf16): 40 KiB, one barrier per K-tile, three workgroups per CU
template <bool A_K, bool B_K>
__global__ __launch_bounds__(256, 3) void ep_gemm_b3_wide1_kernel(GemmParams p, int mtn, int ntn, unsigned ntiles, int xcd) {
  extern __shared__ __attribute__((aligned(1024))) char lds_b3[];
  const unsigned L = blockIdx.x;
  const unsigned V = xcd ? (L % 8u) * (gridDim.x / 8u) + L / 8u : L;
  if (V >= ntiles) return;
  const int bx = __builtin_amdgcn_readfirstlane((int)(V % (unsigned)ntn)), by = __builtin_amdgcn_readfirstlane((int)((V / (unsigned)ntn) % (unsigned)mtn)),
            bz = __builtin_amdgcn_readfirstlane((int)(V / (unsigned)(mtn * ntn)));
  gemm_tile_b3w1<A_K, B_K>(p, bx, by, bz, lds_b3);
}
template <bool A_K, bool B_K>
static void b3_wide_launch(const GemmParams& q3, int batch, hipStream_t st) {
  static bool attr_wide = false;
  if (!attr_wide) { (void)hipFuncSetAttribute((const void*)ep_gemm_b3_wide_kernel<A_K, B_K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3W_LDS_BYTES); attr_wide = true; }
  static int xcd = -1;
  if (xcd < 0) { const char* e = getenv("EP_B3_WIDE_XCD"); xcd = e ? atoi(e) : 1; }
  const int mtn = (q3.M + 127) / 128, ntn = (q3.N + 127) / 128;
  const unsigned ntiles = (unsigned)mtn * (unsigned)ntn * (unsigned)batch;
  static int one_form = -1;                                // EP_B3_WIDE1=0: the three-term tile with its run-time single-term branch
  if (one_form < 0) { const char* e = getenv("EP_B3_WIDE1"); one_form = e ? atoi(e) : 1; }
  if (q3.nterms == 1 && one_form)
    hipLaunchKernelGGL((ep_gemm_b3_wide1_kernel<A_K, B_K>), dim3(8u * ((ntiles + 7u) / 8u)), dim3(256), W3W1_LDS_BYTES, st, q3, mtn, ntn, ntiles, xcd);
  else
    hipLaunchKernelGGL((ep_gemm_b3_wide_kernel<A_K, B_K>), dim3(8u * ((ntiles + 7u) / 8u)), dim3(256), W3W_LDS_BYTES, st, q3, mtn, ntn, ntiles, xcd);
}
static bool b3_wide_ok(const GemmParams& p, bool a_k, bool b_k, int batch) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_GEMM_B3_WIDE"); on = e ? atoi(e) : 1; }
  if (!on || !gemm_b3_on() || p.cs_out || p.ksplit > 1) return false;
  // PER-IMAGE products (>= 64 batch entries of at least 128 x 128 outputs, any layout; round 6, r6.10): the attention products of
  // the AbMILP head -- S = q k^T, O = A v, dA, dv, dq, dk at 256 x 256 x 1152 / 256 x 1152 x 256 per image -- on 64 x 64 tiles were
  // bound by the L2 -> LDS fill of their fp32 operands.  EP_GEMM_B3_WIDE=3: the weight gradients only.
  if (on != 3 && batch >= 64 && p.M >= 128 && p.N >= 128 && p.K >= 64)
    return vec_ok(p.A, p.lda, p.sAz, a_k ? p.K : p.extA) && vec_ok(p.B, p.ldb, p.sBz, b_k ? p.K : p.extB);
  const int floor_ = on == 2 ? 256 : 1024;
  return !a_k && !b_k && !p.bias && p.M >= floor_ && p.N >= floor_ && p.K >= 4096 &&
         vec_ok(p.A, p.lda, p.sAz, p.extA) && vec_ok(p.B, p.ldb, p.sBz, p.extB);
}
// THIN outputs (N <= 32 per batch entry, K / K layout): the value projection at the published protocol's 32 queries is 32 batched
// contractions of 24 / 32 columns (reference poolings/ep.py:40 with --ep_queries 32) -- on 64-column tiles more than half of the
// staging and of the matrix instructions worked on padding (rocprofv3, in the step: 32 - 41 us).  64 x 32 tiles: wave = 32 x 16.
__global__ __launch_bounds__(256) void ep_gemm_b3_thin_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds_b3[];
  gemm_tile_b3g<true, true, 64, 32>(p, blockIdx.x, blockIdx.y, blockIdx.z, lds_b3);
}
constexpr size_t B3_KERNEL_LDS = EP_B3_STAGES * W3_LDS_BYTES;
template <bool A_K, bool B_K>
static void b3_launch(const GemmParams& p, int batch, bool m32, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)ep_gemm_b3_kernel<A_K, B_K, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_KERNEL_LDS);
    (void)hipFuncSetAttribute((const void*)ep_gemm_b3_kernel<A_K, B_K, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_KERNEL_LDS);
    attr_set = true;
  }
  // XCD-grouped order for launches of SEVERAL entries that fill the chip (round 6: the six per-image attention products of the
  // AbMILP step -- 256 x 256 x 1152 per image, 16 - 72 tiles each -- streamed every image's operands into up to eight L2s: 0.27 -
  // 0.55 ms per 39 GFLOP product; AMP-bf16 step at 256 x 1152 8.30 -> 7.95 ms, fp32 16.33 -> 16.15, DINOv2 block 19.8 -> 19.5).
  // EP_B3_XCD=0: launch order everywhere; =2: grouped for every launch.
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("EP_B3_XCD"); mode = e ? atoi(e) : 1; }
  const int gx = (p.N + 63) / 64, gy = m32 ? (p.M + 31) / 32 : (p.M + 63) / 64;
  const unsigned ntiles = (unsigned)gx * (unsigned)gy * (unsigned)batch;
  // (default rule: per-image batches and the K slices of long weight gradients.  The EP step's own batched launches -- 8 / 32 query
  // slices, 1024-row gradients -- measured equal or 0.3 - 1 % slower grouped and keep the launch order.)
  const int xcd = mode == 2 || (mode == 1 && gx * gy > 1 && ntiles >= 2u * (unsigned)cu_count() &&
                                (batch >= 64 || (batch > 1 && !A_K && p.K >= 2048)));
  const unsigned grid = xcd ? 8u * ((ntiles + 7u) / 8u) : ntiles;
  if (m32) hipLaunchKernelGGL((ep_gemm_b3_kernel<A_K, B_K, 32>), dim3(grid), dim3(256), B3_KERNEL_LDS, st, p, gx, gy, ntiles, xcd);
  else hipLaunchKernelGGL((ep_gemm_b3_kernel<A_K, B_K, 64>), dim3(grid), dim3(256), B3_KERNEL_LDS, st, p, gx, gy, ntiles, xcd);
}
// ---- up to two small T / T weight gradients in ONE launch on the paired-group tile (ep_wgrad3.h: gemm_tile_b3p): the EP step's
// dWc = dlogits^T z and dWv_q = dy_q^T P_q as a launch of their own between BatchNorm backward and the second token pass ----
struct PairJobs { GemmParams g[2]; int gx[2], gy[2], nb[2]; };
__global__ __launch_bounds__(512) void ep_wgrad_pair_kernel(PairJobs j) {
  extern __shared__ __attribute__((aligned(1024))) char lds_wp[];
  int L = blockIdx.x, k = 0;
  if (L >= j.nb[0]) { L -= j.nb[0]; k = 1; }
  const int bx = L % j.gx[k], r = L / j.gx[k];
  gemm_tile_b3p(j.g[k], bx, r % j.gy[k], r / j.gy[k], lds_wp);
}
__global__ __launch_bounds__(256) void ep_wgrad_free_kernel(PairJobs j) {
  int L = blockIdx.x, k = 0;
  if (L >= j.nb[0]) { L -= j.nb[0]; k = 1; }
  const int bx = L % j.gx[k], r = L / j.gx[k];
  gemm_tile_b3f(j.g[k], bx, r % j.gy[k], r / j.gy[k]);
}
__global__ __launch_bounds__(256, 2) void ep_wgrad_freek_kernel(PairJobs j) {
  __shared__ __attribute__((aligned(16))) float red[3 * 64 * 16];
  int L = blockIdx.x, k = 0;
  if (L >= j.nb[0]) { L -= j.nb[0]; k = 1; }
  const int bx = L % j.gx[k], r = L / j.gx[k];
  gemm_tile_b3fk(j.g[k], bx, r % j.gy[k], r / j.gy[k], red);
}
bool wgrad_pair_ok(const GemmParams& p) {
  return gemm_b3_on() && !p.bias && !p.cs_out && p.M > 0 && p.N > 0 && p.K >= 64 && vec_ok(p.A, p.lda, p.sAz, p.extA) && vec_ok(p.B, p.ldb, p.sBz, p.extB);
}
int wgrad_pair(const GemmParams* g, const int* batch, int n, hipStream_t st) {
  EP_REQUIRE(n >= 1 && n <= 2, EP_E_ARG, "wgrad_pair: one or two contractions");
  PairJobs j{};
  int total = 0;
  for (int i = 0; i < 2; ++i) {
    if (i < n) {
      EP_REQUIRE(wgrad_pair_ok(g[i]), EP_E_ALIGN, "wgrad_pair: T / T contraction %d needs 16-byte aligned operands, K >= 64, no bias", i);
      j.g[i] = g[i];
      if (gemm_arith() == 1) j.g[i].nterms = 1;
      { static int abl = -1; if (abl < 0) { const char* e = getenv("EP_WG2_ABLATE"); abl = e ? atoi(e) : 0; } j.g[i].ablate = abl; }   // diagnostics: 1 no matrix phase, 2 no staging, 4 no ring refills (results are wrong)
      j.gx[i] = (g[i].N + 63) / 64; j.gy[i] = (g[i].M + 63) / 64; j.nb[i] = j.gx[i] * j.gy[i] * batch[i];
    } else { j.g[i] = g[0]; j.gx[i] = j.gy[i] = 1; j.nb[i] = 0; }
    total += j.nb[i];
  }
  static int form = -1;                              // EP_WG2_FORM: 1 = paired groups through LDS (gemm_tile_b3p), 2 = barrier-free waves
  if (form < 0) { const char* e = getenv("EP_WG2_FORM"); form = e ? atoi(e) : 3; }      // (gemm_tile_b3f), 3 = ... with K dealt over a workgroup's waves (b3fk)
  bool fk_ok = true;                                   // the K-dealt form: whole K-tiles, 32-bit byte offsets
  for (int i = 0; i < n; ++i)
    fk_ok &= g[i].K % 32 == 0 && ((int64_t)g[i].lda * 32 + g[i].M) * 4 < (int64_t)0x7fffffff && ((int64_t)g[i].ldb * 32 + g[i].N) * 4 < (int64_t)0x7fffffff;
  if (form == 3 && !fk_ok) { hipLaunchKernelGGL(ep_wgrad_free_kernel, dim3(total), dim3(256), 0, st, j); EP_LAUNCH_CHECK("ep_wgrad_free_kernel"); return 0; }
  if (form == 3) {
    PairJobs q = j;
    int tot = 0;
    for (int i = 0; i < 2; ++i) {
      if (i < n) { q.gx[i] = (g[i].N + 31) / 32; q.gy[i] = (g[i].M + 31) / 32; q.nb[i] = q.gx[i] * q.gy[i] * batch[i]; }
      tot += q.nb[i];
    }
    hipLaunchKernelGGL(ep_wgrad_freek_kernel, dim3(tot), dim3(256), 0, st, q);
    EP_LAUNCH_CHECK("ep_wgrad_freek_kernel");
    return 0;
  }
  if (form == 2) {
    hipLaunchKernelGGL(ep_wgrad_free_kernel, dim3(total), dim3(256), 0, st, j);
    EP_LAUNCH_CHECK("ep_wgrad_free_kernel");
    return 0;
  }
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)ep_wgrad_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3P_LDS_BYTES); attr = true; }
  hipLaunchKernelGGL(ep_wgrad_pair_kernel, dim3(total), dim3(512), W3P_LDS_BYTES, st, j);
  EP_LAUNCH_CHECK("ep_wgrad_pair_kernel");
  return 0;
}
static thread_local int t_arith = 0;
int gemm_arith() { return t_arith; }
void gemm_set_arith(int a) { t_arith = a; }
// EP_GEMM_B3=0: T / T contractions back on the exact-f32 kernels
bool gemm_b3_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_GEMM_B3"); on = e ? atoi(e) : 1; }
  return on != 0;
}
// T / T (weight gradients): always.  The other layouts (activation x weight): LARGE contractions only -- the 1024-row
// critical-path ones of the EP step are one tile per CU and latency-bound, where the LDS-DMA ring kernels stay ahead
// (EP_GEMM_B3_MIN_TILES: 64 x 64 tiles from which the bf16 x3 tile takes a K / K or K / T contraction; 0 = never).
bool gemm_b3_ok(const GemmParams& p, bool a_k, bool b_k, int batch) {
  if (!gemm_b3_on() || p.K < 64 || p.cs_out) return false;
  if (!vec_ok(p.A, p.lda, p.sAz, a_k ? p.K : p.extA) || !vec_ok(p.B, p.ldb, p.sBz, b_k ? p.K : p.extB)) return false;
  if (!a_k && !b_k) {
    if (p.bias) return false;
    if (t_arith == 1 || p.nterms == 1) return true;            // AMP-bf16: this tile is the single-product kernel of the T / T layout
    // `side`: the contraction runs on a second queue BESIDE a token pass (shapes whose pass cannot carry it as side
    // workgroups).  The tile's split instructions then compete with a vector-issue-bound stream, where the exact-f32
    // kernel only uses the otherwise idle matrix pipe: 196 x 1024 0.536 against 0.522 ms, 256 x 1152 0.733 against 0.726 ms
    // with it -- but 196 x 4096 (34 GFLOP of dWv beside the dP contraction, not beside a pass) 2.284 against 2.335 ms.
    // So: side contractions only from 8 GFLOP.
    static double side_min = -1.0;                             // (EP_GEMM_B3_SIDE_MIN_GFLOP: experiments)
    if (side_min < 0) { const char* e = getenv("EP_GEMM_B3_SIDE_MIN_GFLOP"); side_min = e ? atof(e) * 1e9 : 8e9; }
    if (p.side && 2.0 * p.M * p.N * (double)p.K * batch < side_min) return false;
    return true;
  }
  if (!a_k) return false;                                      // (T / K does not occur)
  // AMP-bf16 (round 6: the CoCa / AbMILP steps and the EP step's dP at odd slice widths): this tile is the single-product
  // kernel of the K / K and K / T layouts too, whatever the size
  if (t_arith == 1 || p.nterms == 1) return true;
  static long min_tiles = -1;
  if (min_tiles < 0) { const char* e = getenv("EP_GEMM_B3_MIN_TILES"); min_tiles = e ? atol(e) : 512; }
  const long tiles64 = (long)((p.N + BN - 1) / BN) * ((p.M + 63) / 64) * batch;
  // K floor 128 since round 6 (EP_GEMM_B3_MIN_K; 256 before): dP = dy_q Wv_q at 256 x 1152 (K = 144, 2304 tiles) 79.9 -> 37.4 us
  // (rocprofv3); the step gains less -- 0.764 -> 0.759 ms, bf16-stored tokens 0.507 -> 0.499 -- because the weight gradients of
  // the second queue then overlap the second token pass for longer (277 -> 311 us).  Every other head and shape: equal.
  static int min_k = -1;
  if (min_k < 0) { const char* e = getenv("EP_GEMM_B3_MIN_K"); min_k = e ? atoi(e) : 128; }
  return min_tiles > 0 && tiles64 >= min_tiles && p.K >= min_k;
}

const char* gemm_kernel_name(bool a_k, bool b_k, const GemmParams& p, int batch) {
  if (gemm_b3_ok(p, a_k, b_k, batch)) return "ep_gemm_b3_kernel";
  if (gemm_kk96_ok(a_k, b_k, p, batch)) return "ep_gemm_kk96_kernel";
  if (gemm_kt96_ok(a_k, b_k, p, batch)) return "ep_gemm_kt96_kernel";
  return "ep_gemm_dma_kernel";
}

int gemm(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  if (batch == 1 && p.skws) {
    bool done = false;
    EP_TRY(gemm_split_k(a_k, b_k, p, st, &done));
    if (done) return 0;
  }
  const bool vec = vec_ok(p.A, p.lda, p.sAz, a_k ? p.K : p.extA) && vec_ok(p.B, p.ldb, p.sBz, b_k ? p.K : p.extB);
  const long tiles64 = (long)((p.N + BN - 1) / BN) * ((p.M + 63) / 64) * batch;
  if (gemm_b3_ok(p, a_k, b_k, batch)) {
    // 32-row tiles where 64-row ones would leave tiles half empty or the chip under-filled (as side_add_gemm)
    const bool m32 = !(p.M % 64 == 0 || p.M >= 256) || tiles64 < 2L * cu_count();
    GemmParams q3 = p;
    if (t_arith == 1) q3.nterms = 1;
    static int thin_on = -1;                                   // EP_GEMM_B3_THIN=0: 64-column tiles for thin outputs too
    if (thin_on < 0) { const char* e = getenv("EP_GEMM_B3_THIN"); thin_on = e ? atoi(e) : 1; }
    if (b3_wide_ok(p, a_k, b_k, batch) && (a_k || !b_k)) {               // (T / K does not occur)
      if (!a_k) b3_wide_launch<false, false>(q3, batch, st);
      else if (b_k) b3_wide_launch<true, true>(q3, batch, st);
      else b3_wide_launch<true, false>(q3, batch, st);
      EP_LAUNCH_CHECK("ep_gemm_b3_wide_kernel");
      return 0;
    }
    if (thin_on && a_k && b_k && p.N <= 32 && p.M >= 64) {
      static bool attr_thin = false;
      if (!attr_thin) { (void)hipFuncSetAttribute((const void*)ep_gemm_b3_thin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS_BYTES); attr_thin = true; }
      hipLaunchKernelGGL(ep_gemm_b3_thin_kernel, dim3((p.N + 31) / 32, (p.M + 63) / 64, batch), dim3(256), W3_LDS_BYTES, st, q3);
      EP_LAUNCH_CHECK("ep_gemm_b3_thin_kernel");
      return 0;
    }
    if (!a_k && p.skws && EP_B3_STAGES == 1) {
      // SMALL weight gradients (the 1024-row gradients of the EP step where they run on side queues: 192 - 384 tiles, one or two
      // workgroups per CU, every K-tile a latency chain with nothing beside it -- 43 / 61 us for dWc / dWv at 32 queries,
      // EXPERIMENTS.md r6.1): K slices as extra batch entries of ONE launch, so that ~3 workgroups share a CU, summed in slice
      // order by ep_reduce_partials_kernel (deterministic; on the side queue the extra launch is off the critical path).
      // EP_B3_SPLITK_WGS: workgroups per CU aimed at (0 = off, the DEFAULT: measured neutral -- same-box, ms per step, off / 3 / 4:
      // 196 x 1024 Q = 32 0.776 / 0.785 / 0.785, 256 x 768 Q = 32 bf16 0.4237 / 0.4256 / 0.430, fp32 0.708 / 0.705 / 0.711, 256 x 1152
      // Q = 32 1.265 / 1.257 / 1.263: beside the contractions of the main queue the gradients are bound by what they share with
      // them, not by their own occupancy.  The scratch is only carved with the switch on.)
      static int target = -1;
      if (target < 0) { const char* e = getenv("EP_B3_SPLITK_WGS"); target = e ? atoi(e) : 0; }
      const long cus = cu_count();
      const long tiles = (long)((p.N + 63) / 64) * ((p.M + (m32 ? 31 : 63)) / (m32 ? 32 : 64)) * batch;
      const int64_t mn = (int64_t)p.M * p.N;
      int S = 1;
      if (target > 0 && tiles < target * cus && p.ldc == p.N && p.alpha == 1.0f && !p.bias && (batch == 1 || p.sCz == mn) && mn % 4 == 0 &&
          (int64_t)batch * mn < (int64_t)0x7fffffff)
        for (int sp = 2; sp <= 8; sp *= 2) {
          if (p.K % (sp * 32) != 0 || p.K / sp < 128 || (size_t)sp * batch * mn > p.skws_floats) break;
          S = sp;
          if (tiles * sp >= target * cus) break;
        }
      if (S > 1) {
        q3.K = p.K / S; q3.ksplit = S; q3.ksA = (int64_t)q3.K * p.lda; q3.ksB = (int64_t)q3.K * p.ldb;
        q3.C = p.skws; q3.sCz = mn; q3.ksC = (int64_t)batch * mn; q3.accumulate = 0; q3.skws = nullptr;
        b3_launch<false, false>(q3, batch * S, m32, st);
        EP_LAUNCH_CHECK("ep_gemm_b3_kernel (K slices)");
        return reduce_partials(p.skws, S, (int)(batch * mn), 1.0f, p.accumulate, p.C, nullptr, st);
      }
    }
    if (!a_k) b3_launch<false, false>(q3, batch, m32, st);
    else if (b_k) b3_launch<true, true>(q3, batch, m32, st);
    else b3_launch<true, false>(q3, batch, m32, st);
    EP_LAUNCH_CHECK("ep_gemm_b3_kernel");
    return 0;
  }
  static int force_bm = -1;
  if (force_bm < 0) { const char* e = getenv("EP_GEMM_BM"); force_bm = e ? atoi(e) : 0; }
  static int use_ws = -1;
  if (use_ws < 0) { const char* e = getenv("EP_GEMM_WS"); use_ws = e ? atoi(e) : 1; }
  const bool small = force_bm ? (force_bm == 32) : (tiles64 < 2L * cu_count());
  // wave-specialised kernel: pays off for long K on the critical path; bit 1 of EP_GEMM_WS also enables it
  // for the weight-gradient contractions that run beside the second token pass
  const bool ws_ok = use_ws && vec && !force_bm && p.K >= 256 && (!p.side || (use_ws & 2));
  static int use_dma = -1;
  if (use_dma < 0) { const char* e = getenv("EP_GEMM_DMA"); use_dma = e ? atoi(e) : 1; }
  if (use_dma && vec && !force_bm && gemm_kk96_ok(a_k, b_k, p, batch)) {
    hipLaunchKernelGGL(ep_gemm_kk96_kernel, dim3(p.N / 96, (p.M + 31) / 32, batch), dim3(512), 0, st, p);
  } else if (use_dma && vec && !force_bm && gemm_kt96_ok(a_k, b_k, p, batch)) {
    hipLaunchKernelGGL(ep_gemm_kt96_kernel, dim3(p.N / 96, (p.M + 31) / 32, batch), dim3(512), 0, st, p);
  } else if (use_dma && vec && !force_bm) {
    // short K and many tiles: every workgroup walks several N-tiles through one ring (see the kernel) -- as many as still
    // leave one workgroup per CU.  dP = dy_q Wv_q at 1024 x 768, Q = 8 (K = 96, 1536 tiles): 22.7 us with one tile per
    // workgroup, 20.8 / 19.6 / 19.1 us with 2 / 3 / 6.  EP_GEMM_NPERS: 1 = off, n > 1 = at most n.
    GemmParams q = p;
    static int npers_env = -1;
    if (npers_env < 0) { const char* e = getenv("EP_GEMM_NPERS"); npers_env = e ? atoi(e) : 0; }
    const int ntn = (p.N + BN - 1) / BN, nk = (p.K + BK - 1) / BK;
    if (p.N % BN == 0 && p.K % BK == 0 && nk <= 8 && ntn > 1 && tiles64 >= 4L * cu_count()) {
      int np = npers_env > 0 ? npers_env : ntn;
      if (np > ntn) np = ntn;
      while (np > 1 && (ntn % np != 0 || tiles64 / np < cu_count())) --np;
      q.npers = np;
    }
    gemm_launch_dma(a_k, b_k, q, batch, st);
  }
  else if (ws_ok) gemm_launch_ws(a_k, b_k, p, batch, st);
  else if (small) gemm_launch<32>(a_k, b_k, vec, p, batch, st);
  else gemm_launch<64>(a_k, b_k, vec, p, batch, st);
  EP_LAUNCH_CHECK("ep_gemm_kernel");
  return 0;
}

}  // namespace ep
