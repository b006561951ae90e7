// LDS-DMA tile of the exact-fp32 contractions (see ep_gemm.hip for the design notes): shared by the stand-alone kernel
// and by the side tasks of the second token pass.
#pragma once
#include "ep_side.h"

namespace ep {

__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int N>
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the tile body as a device function of the tile coordinates: the stand-alone kernel below and the side tasks of the second
// token pass (ep_side.h: NST = 3, !WS, 256 threads, the pass's own LDS) share it.  `lds`: NST * 16 KiB, 1 KiB aligned.
template <bool A_K, bool B_K, int NST, bool WS>
__device__ __forceinline__ void gemm_dma_tile(const GemmParams& p, int bx_, int by_, int bz_, char* lds) {
  constexpr int OPB = 64 * BK * 4;                   // bytes per operand image (8 KiB)
  constexpr int STB = 2 * OPB;                       // bytes per ring stage
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = WS && wall >= 4;               // WS: waves 4-7 only move data, waves 0-3 only multiply
  const int w = wall & 3;
  const int wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, kk = lane >> 4;
  // npers > 1 (host: N % 64 == 0, K % 32 == 0): the workgroup walks npers consecutive N-tiles of its row block through ONE
  // ring -- the K-tiles of N-tile t+1 follow those of N-tile t without a drain, so the ~5 us a launch-sized pipeline spends
  // filling and emptying is paid once per workgroup instead of once per 64x64 tile (contractions with a short K and many
  // tiles: dP = dy_q Wv_q with K = 96)
  const int npers = p.npers > 1 ? p.npers : 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int nt0 = bx_ * npers;
  const int my_nt = (ntn - nt0) < npers ? (ntn - nt0) : npers;
  const int m0 = by_ * 64, n0 = nt0 * BN;
  const int z = bz_;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int nk = (p.K + BK - 1) / BK;
  const int total = my_nt * nk;                      // ring steps of this workgroup

  // ---- DMA source offsets (elements) of this lane's two pieces per operand, for a full K-tile at k0 = 0 ----
  // piece pc = w + 4 jj covers LDS positions pc*64 + lane
  int64_t srcA[2], srcB[2];
  int64_t kstepA, kstepB;
  auto src_off = [&](bool klay, int64_t ld, int lim, int ext, int r0, int pos, int k0) -> int64_t {
    // lim: number of rows (K layout) ; ext: readable extent along the contiguous dim (T layout)
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
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int pos = (w + 4 * jj) * 64 + lane;
    srcA[jj] = src_off(A_K, p.lda, p.M, p.extA, m0, pos, 0);
    srcB[jj] = src_off(B_K, p.ldb, p.N, p.extB, n0, pos, 0);
  }
  kstepA = A_K ? BK : (int64_t)BK * p.lda;
  kstepB = B_K ? BK : (int64_t)BK * p.ldb;
  const bool ktail = (p.K % BK) != 0;
  const int64_t ntstepB = B_K ? (int64_t)BN * p.ldb : (int64_t)BN;   // one N-tile further along B
  int is_t = 0, is_kt = 0, is_nt = 0, is_stage = 0;   // the issuing wave's position in the (N-tile, K-tile) sequence
  auto issue = [&](int) {                            // DMA the next tile of the sequence (clamped to the last one) into the next stage
    const int tt = is_kt;
    char* st = lds + is_stage * STB;
    const int64_t nb = is_nt * ntstepB;
    if (ktail && tt == nk - 1) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int pos = (w + 4 * jj) * 64 + lane;
        const int64_t oa = src_off(A_K, p.lda, p.M, p.extA, m0, pos, tt * BK);
        const int64_t ob = src_off(B_K, p.ldb, p.N, p.extB, n0, pos, tt * BK);
        __builtin_amdgcn_global_load_lds((gptr_t)(A + oa), (lds_ptr_t)(st + (w + 4 * jj) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(B + ob), (lds_ptr_t)(st + OPB + (w + 4 * jj) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        __builtin_amdgcn_global_load_lds((gptr_t)(A + srcA[jj] + tt * kstepA), (lds_ptr_t)(st + (w + 4 * jj) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(B + srcB[jj] + nb + tt * kstepB), (lds_ptr_t)(st + OPB + (w + 4 * jj) * 1024), 16, 0, 0);
      }
    }
    is_stage = (is_stage + 1 == NST) ? 0 : is_stage + 1;
    if (is_t < total - 1) {                          // past the end: the last tile again (keeps the vmcnt arithmetic uniform)
      ++is_t;
      if (++is_kt == nk) { is_kt = 0; ++is_nt; }
    }
  };

  if (WS && loader) {
    // loader waves: tiles 0 .. NST-2 in flight, then one K-tile per step behind the same barriers as the matrix waves
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t);
    dma_wait<4 * (NST - 2)>();
    ws_barrier();                                    // tile 0 landed
    for (int it = 0; it < total; ++it) {
      dma_wait<4 * (NST - 3 >= 0 ? NST - 3 : 0)>();  // tile it+1 landed (this wave's pieces)
      ws_barrier();                                  // ... everyone's; the matrix waves hold tile `it` in registers
      issue(it + NST - 1);                           // refill the stage tile it-1 lived in
    }
    dma_wait<0>();
    return;
  }

  // ---- fragment addressing (bytes inside an operand image) ----
  // K layout: block mi, group g : row r = base + 16 mi + i16 ; chunk 4g + kk -> r*128 + 16*((4g+kk) ^ ((r>>1)&7))
  // T layout: value (g, j)      : k = 16g + 4kk + j ; col = base + 16 mi + i16
  //           -> (16g + j)*256 + [4kk*256 + 16*((col>>2) ^ 4(kk&1)) + 4(col&3)]
  int fragA[2][2], fragB[2][2];                      // [block][g] (K layout) or [block][0] (T layout: lane base)
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
    {
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
      const int r = wn * 32 + bi * 16 + i16;
      if (B_K) {
        fragB[bi][0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
        fragB[bi][1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
      } else {
        fragB[bi][0] = 4 * kk * 256 + 16 * ((r >> 2) ^ (4 * (kk & 1))) + 4 * (r & 3);
        fragB[bi][1] = 0;
      }
    }
  }
  // fragment registers: [set][block][g] as f4 (elements j = 0..3)
  f4v fa[2][2][2], fb[2][2][2];
  auto read_frags = [&](int stage, f4v (&xa)[2][2], f4v (&xb)[2][2]) {
    const char* sa = lds + stage * STB;
    const char* sb = sa + OPB;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (A_K) xa[bi][g] = *reinterpret_cast<const f4v*>(sa + fragA[bi][g]);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) xa[bi][g][j] = *reinterpret_cast<const float*>(sa + fragA[bi][0] + (16 * g + j) * 256);
        }
        if (B_K) xb[bi][g] = *reinterpret_cast<const f4v*>(sb + fragB[bi][g]);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) xb[bi][g][j] = *reinterpret_cast<const float*>(sb + fragB[bi][0] + (16 * g + j) * 256);
        }
      }
  };
  f4v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const f4v (&xa)[2][2], const f4v (&xb)[2][2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[mi][g][j], xb[ni][g][j], acc[mi][ni], 0, 0, 0);
  };
  auto zero_tail = [&](int k0, f4v (&xa)[2][2], f4v (&xb)[2][2]) {   // last tile only: k >= K contributes nothing
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool out = k0 + 16 * g + 4 * kk + j >= p.K;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
          xa[bi][g][j] = out ? 0.f : xa[bi][g][j];
          xb[bi][g][j] = out ? 0.f : xb[bi][g][j];
        }
      }
  };

  // ---- pipeline ----
  // prologue: tiles 0 .. NST-2 in flight; tile 0 landed -> fragments of tile 0 in set 0
  if (!WS) {
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t);
    dma_wait<4 * (NST - 2)>();                       // this wave's pieces of tile 0 (4 DMA instructions per tile)
  }
  ws_barrier();                                      // ... and every other wave's
  read_frags(0, fa[0], fb[0]);
  if (ktail && nk == 1) { __builtin_amdgcn_s_waitcnt(0xc07f); zero_tail(0, fa[0], fb[0]); }
  // step it (set F = it % 2): tile it+1 landed (vmcnt + barrier; the barrier also says every wave has tile `it` in
  // registers, so its stage can be refilled) -> DMA tile it+NST-1 into that stage, read the fragments of tile it+1
  // into the other set, multiply tile it.
#define EP_DMA_STEP(IT, F)                                                         \
  {                                                                                \
    if (!WS) dma_wait<4 * (NST - 3 >= 0 ? NST - 3 : 0)>();                         \
    __builtin_amdgcn_s_waitcnt(0xc07f);                                            \
    ws_barrier();                                                                  \
    if (!WS) issue((IT) + NST - 1);                                                \
    read_frags(((IT) + 1) % NST, fa[(F) ^ 1], fb[(F) ^ 1]);                        \
    __builtin_amdgcn_sched_barrier(0);                                             \
    multiply(fa[F], fb[F]);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                             \
    if (ktail && (IT) + 1 == nk - 1) { __builtin_amdgcn_s_waitcnt(0xc07f); zero_tail(((IT) + 1) * BK, fa[(F) ^ 1], fb[(F) ^ 1]); } \
  }
  auto store_tile = [&](int ncol0) {
    f4v blk[4]; int rb[4], cb[4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + wm * 32 + mi * 16; cb[mi * 2 + ni] = ncol0 + wn * 32 + ni * 16;
        acc[mi][ni] = f4v{0.f, 0.f, 0.f, 0.f};
      }
    store_acc_blocks<4>(p, C, z, rb, cb, blk, kk, i16);
  };
  int ckt = 0, cn0 = n0;                             // K-tile inside the current N-tile, its first column
  auto tile_end = [&]() {
    if (++ckt == nk) { ckt = 0; store_tile(cn0); cn0 += BN; }
  };
  int it = 0;
  for (; it + 1 < total; it += 2) {
    EP_DMA_STEP(it, 0)
    tile_end();
    EP_DMA_STEP(it + 1, 1)
    tile_end();
  }
  if (it < total) { EP_DMA_STEP(it, 0) tile_end(); }
#undef EP_DMA_STEP
  if (!WS) dma_wait<0>();                            // redundant prefetches past the last tile: drain before exit
}


}  // namespace ep
