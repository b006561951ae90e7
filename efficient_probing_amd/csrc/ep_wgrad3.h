// Contraction tile on the bf16 matrix cores at fp32 accuracy with BOTH operands split on the fly (gfx950).  Built for the
// weight gradients:
//
//     C[m][n] (+)= alpha * sum_k A[k][m] * B[k][n]          (both operands "T layout": k = the batch index is the slow one)
//
// (gemm_tile_b3g takes K-contiguous operands as well -- gemm() hands it the K / K and K / T contractions of 512 output tiles
// and more, where it measured 1 - 1.5 % on the steps of the attention-pool and matrix-core-bound heads.)
//
// -- dWc = dlogits^T z, dWv_q = dy_q^T P_q and every other sum-over-the-batch gradient of a Linear (reference
// probe_heads.py:76, poolings/ep.py:40 under autograd).  BOTH operands are fp32 activations, so unlike ep_planes.hip (weights
// pre-split once per step) both are split HERE, once per workgroup and K-tile, on their way from registers into LDS:
//
//   x = h + m + l exactly (three bf16 terms, round-to-nearest: ep_planes_dev.h pl_split2), and
//   a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [<= 2^-24 |a*b|, dropped]
//
// i.e. six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block at 16 matrix cycles each instead of eight
// v_mfma_f32_16x16x4_f32 at 32: 96 against 256 matrix cycles.  The f32 tile (ep_side.h: gemm_tile) spends 1024 matrix cycles
// per wave and K-tile and is bound by them -- 544 side tiles of the EP step at 256 x 768 are 27 us of matrix time on the
// whole chip and ~45 us in the tail of the bf16-token second pass.
//
// A K-tile is 32 batch rows.  A thread fetches a k-PAIR x 4 consecutive m (two float4, coalesced along m), splits the four
// (x[2kp][m], x[2kp+1][m]) pairs into packed bf16 terms -- one b32 holds the two k of a pair, which is how the matrix
// instruction wants them: 8 consecutive k per lane -- and writes them TRANSPOSED into plane images [term][m][k]:
//   row stride 80 bytes (64 of data), 16-byte k-chunk c of row r stored at chunk c ^ ((r >> 4) & 3):
//   writes (16 lanes on rows 4 mq + j) and fragment reads (ds_read_b128 of 16 consecutive rows) are both conflict-free.
// One LDS stage (two operands x three terms x 5 KiB = 30 KiB), the next K-tile's global loads in flight in registers
// while this one is multiplied; two barriers per K-tile (the 2 - 3 resident workgroups of a CU cover them).
// Wave (wm, wn) of the 2 x 2 owns BMT/2 x 32 of the BMT x 64 output tile.
#pragma once
#include "ep_side.h"
#include "ep_planes_dev.h"

namespace ep {

constexpr int W3_ROWB = 80;                       // bytes per plane-image row (32 k x 2 B + 16 B pad)
constexpr int W3_IMG = 64 * W3_ROWB;              // one term of one operand (64 rows)
constexpr size_t W3_LDS_BYTES = 6 * (size_t)W3_IMG;      // A: h m l | B: h m l  = 30720

__device__ __forceinline__ int w3_off(int row, int kchunk) { return row * W3_ROWB + ((kchunk ^ ((row >> 4) & 3)) << 4); }

// ---- T layout (k is the slow index): a thread takes a k-PAIR x 4 consecutive columns ------------------------------------
// fp32 tile rows k0 + 2 kp, + 1, columns c0 + 4 mq .. + 3 -> registers (branch-free; masked in w3_stage_T)
__device__ __forceinline__ void w3_load_T(const float* __restrict__ base, int64_t ld, int ext, int K, int c0, int k0, int kp, int mq,
                                          f4v (&x)[2]) {
  const int c = c0 + 4 * mq;
  const int cc = c < ext ? c : 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = k0 + 2 * kp + h;
    x[h] = *reinterpret_cast<const f4v*>(base + (int64_t)(k < K ? k : K - 1) * ld + cc);
  }
}
// split and store this thread's 2 x 4 values into the three plane images of one operand (transposed: image rows = columns)
__device__ __forceinline__ void w3_stage_T(char* img, const f4v (&x)[2], int ext, int K, int c0, int k0, int kp, int mq, bool one = false,
                                           int tstride = W3_IMG) {   // tstride: bytes between the three term images (128-row images: 2 W3_IMG)
  const bool cok = c0 + 4 * mq < ext;
  const bool k0ok = cok && k0 + 2 * kp < K, k1ok = cok && k0 + 2 * kp + 1 < K;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, m, l;
    pl_split2(k0ok ? x[0][j] : 0.f, k1ok ? x[1][j] : 0.f, h, m, l);
    const int o = w3_off(4 * mq + j, kp >> 2) + 4 * (kp & 3);
    *reinterpret_cast<unsigned*>(img + o) = h;
    if (one) continue;                               // AMP-bf16 (GemmParams.nterms == 1): the rounded operand only
    *reinterpret_cast<unsigned*>(img + tstride + o) = m;
    *reinterpret_cast<unsigned*>(img + 2 * tstride + o) = l;
  }
}
// ---- K layout (k contiguous): a thread takes 4 consecutive k of one row (two rows for 64-row tiles) ----------------------
template <int RT>
__device__ __forceinline__ void w3_load_K(const float* __restrict__ base, int64_t ld, int rows, int K, int r0, int k0, int tid,
                                          f4v (&x)[2]) {
#pragma unroll
  for (int r = 0; r < RT / 32; ++r) {
    const int idx = tid + 256 * r;
    const int row = r0 + (idx >> 3), k = k0 + 4 * (idx & 7);
    x[r] = *reinterpret_cast<const f4v*>(base + (int64_t)(row < rows ? row : rows - 1) * ld + (k < K ? k : 0));
  }
}
template <int RT>
__device__ __forceinline__ void w3_stage_K(char* img, const f4v (&x)[2], int rows, int K, int r0, int k0, int tid, bool one = false,
                                           int tstride = W3_IMG) {
#pragma unroll
  for (int r = 0; r < RT / 32; ++r) {
    const int idx = tid + 256 * r, lrow = idx >> 3, kq = idx & 7;
    const bool ok = r0 + lrow < rows && k0 + 4 * kq < K;
    unsigned h0, m0, l0, h1, m1, l1;
    pl_split2(ok ? x[r][0] : 0.f, ok ? x[r][1] : 0.f, h0, m0, l0);
    pl_split2(ok ? x[r][2] : 0.f, ok ? x[r][3] : 0.f, h1, m1, l1);
    typedef unsigned w3_u2 __attribute__((ext_vector_type(2)));
    const int o = w3_off(lrow, kq >> 1) + 8 * (kq & 1);
    *reinterpret_cast<w3_u2*>(img + o) = w3_u2{h0, h1};
    if (one) continue;
    *reinterpret_cast<w3_u2*>(img + tstride + o) = w3_u2{m0, m1};
    *reinterpret_cast<w3_u2*>(img + 2 * tstride + o) = w3_u2{l0, l1};
  }
}

// C (+)= alpha * op(A) op(B)^T (+ bias) on one BMT x 64 tile.  A_K / B_K: the operand is contiguous along K (true) or along
// its free dimension (false), as in ep_side.h: gemm_tile.  BMT = 64: 2 x 2 blocks per wave; 32: 1 x 2.  16-byte aligned
// operands with leading dimensions (and K, for K-layout operands) multiples of 4 only.
template <bool A_K, bool B_K, int BMT, int BNT = 64>
__device__ __forceinline__ void gemm_tile_b3g(const GemmParams& p, int bx, int by, int bz, char* lds) {
  constexpr int MI = BMT / 32, NI = BNT / 32;        // 16 x 16 blocks per wave along M / N (BNT = 32: thin outputs, e.g. a 24-column query slice)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * BMT, n0 = bx * BNT;
  int zb = bz, ks = 0;                               // (GemmParams.ksplit: K slices as extra batch entries)
  if (p.ksplit > 1) { zb = bz / p.ksplit; ks = bz - zb * p.ksplit; }
  const float* A = p.A + (int64_t)zb * p.sAz + (int64_t)ks * p.ksA;
  const float* B = p.B + (int64_t)zb * p.sBz + (int64_t)ks * p.ksB;
  float* C = p.C + (int64_t)zb * p.sCz + (int64_t)ks * p.ksC;
  const int i16 = lane & 15, kk = lane >> 4;
  char* imgA = lds;
  char* imgB = lds + 3 * W3_IMG;
  // T-layout staging role: k-pair kp (0..15) x column quad mq (0..15); lanes run over mq first (coalesced rows)
  const int mq = tid & 15, kp = tid >> 4;
  const bool stA = A_K || 4 * mq < BMT;             // (T layout, 32-row tiles: half of the threads have no A work)
  const bool stB = B_K || 4 * mq < BNT;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;

  f4v acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + 31) / 32;
  const bool one = p.nterms == 1;
  f4v xa[2], xb[2];
  auto loadA = [&](int k0) {
    if constexpr (A_K) w3_load_K<BMT>(A, p.lda, p.M, p.K, m0, k0, tid, xa);
    else w3_load_T(A, p.lda, extA, p.K, m0, k0, kp, stA ? mq : 0, xa);
  };
  auto loadB = [&](int k0) {
    if constexpr (B_K) w3_load_K<BNT>(B, p.ldb, p.N, p.K, n0, k0, tid, xb);
    else w3_load_T(B, p.ldb, extB, p.K, n0, k0, kp, stB ? mq : 0, xb);
  };
  loadA(0); loadB(0);
  for (int it = 0; it < nk; ++it) {
    if (it > 0) __syncthreads();                     // every wave has read tile it-1's fragments
    if constexpr (A_K) w3_stage_K<BMT>(imgA, xa, p.M, p.K, m0, it * 32, tid, one);
    else { if (stA) w3_stage_T(imgA, xa, extA, p.K, m0, it * 32, kp, mq, one); }
    if constexpr (B_K) w3_stage_K<BNT>(imgB, xb, p.N, p.K, n0, it * 32, tid, one);
    else { if (stB) w3_stage_T(imgB, xb, extB, p.K, n0, it * 32, kp, mq, one); }
    if (it + 1 < nk) { loadA((it + 1) * 32); loadB((it + 1) * 32); }     // in flight while this tile is multiplied
    __syncthreads();                                 // the plane images are complete
    pl_u4 fa[MI][3], fb[NI][3];
    if (one) {                                       // AMP-bf16: bf16(a) x bf16(b), fp32 accumulation -- one product
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi][0] = *reinterpret_cast<const pl_u4*>(imgA + w3_off(wm * (16 * MI) + mi * 16 + i16, kk));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni][0] = *reinterpret_cast<const pl_u4*>(imgB + w3_off(wn * (16 * NI) + ni * 16 + i16, kk));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = pl_mfma(fa[mi][0], fb[ni][0], acc[mi][ni]);
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fa[mi][t] = *reinterpret_cast<const pl_u4*>(imgA + t * W3_IMG + w3_off(wm * (16 * MI) + mi * 16 + i16, kk));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fb[ni][t] = *reinterpret_cast<const pl_u4*>(imgB + t * W3_IMG + w3_off(wn * (16 * NI) + ni * 16 + i16, kk));
    // smallest terms first: lo x hi, hi x lo, mid x mid, then the 2^-8 pair, then hi x hi (as ep_planes.hip)
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = pl_mfma(fa[mi][ta], fb[ni][tb], acc[mi][ni]);
    }
  }
  f4v blk[MI * NI]; int rb[MI * NI], cb[MI * NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      blk[mi * NI + ni] = acc[mi][ni]; rb[mi * NI + ni] = m0 + wm * (16 * MI) + mi * 16; cb[mi * NI + ni] = n0 + wn * (16 * NI) + ni * 16;
    }
  store_acc_blocks<MI * NI>(p, C, zb, rb, cb, blk, kk, i16);
  __syncthreads();                                   // LDS free for the caller's next tile
}

// The stand-alone form: TWO LDS stages (60 KiB) and two register sets.  Per K-tile ONE barrier; between it and the next a wave
// reads tile it's fragments, then splits and stages tile it+1 (whose rows were fetched two tiles ahead) into the other
// stage -- its ~110 vector instructions cover the fragment reads' latency -- and then issues tile it's matrix instructions
// while the other resident wave of the SIMD is in its staging phase.  (The side-task form above keeps one stage: it has to
// fit the LDS of the token pass it rides in.)
template <bool A_K, bool B_K, int BMT>
__device__ __forceinline__ void gemm_tile_b3g2(const GemmParams& p, int bx, int by, int bz, char* lds) {
  constexpr int MI = BMT / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * BMT, n0 = bx * 64;
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  const int mq = tid & 15, kp = tid >> 4;
  const bool stA = A_K || 4 * mq < BMT;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  f4v acc[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  const int nk = (p.K + 31) / 32;
  f4v xa[2][2], xb[2][2];                            // [register set][...]
  auto load = [&](int t, f4v (&ra)[2], f4v (&rb)[2]) {
    const int k0 = (t < nk ? t : nk - 1) * 32;       // (clamped: redundant, never out of range)
    if constexpr (A_K) w3_load_K<BMT>(A, p.lda, p.M, p.K, m0, k0, tid, ra);
    else w3_load_T(A, p.lda, extA, p.K, m0, k0, kp, stA ? mq : 0, ra);
    if constexpr (B_K) w3_load_K<64>(B, p.ldb, p.N, p.K, n0, k0, tid, rb);
    else w3_load_T(B, p.ldb, extB, p.K, n0, k0, kp, mq, rb);
  };
  auto stage = [&](int t, const f4v (&ra)[2], const f4v (&rb)[2]) {
    char* imgA = lds + (t & 1) * (int)W3_LDS_BYTES;
    char* imgB = imgA + 3 * W3_IMG;
    if constexpr (A_K) w3_stage_K<BMT>(imgA, ra, p.M, p.K, m0, t * 32, tid);
    else { if (stA) w3_stage_T(imgA, ra, extA, p.K, m0, t * 32, kp, mq); }
    if constexpr (B_K) w3_stage_K<64>(imgB, rb, p.N, p.K, n0, t * 32, tid);
    else w3_stage_T(imgB, rb, extB, p.K, n0, t * 32, kp, mq);
  };
  auto step = [&](int it, f4v (&ra_cur)[2], f4v (&rb_cur)[2], f4v (&ra_nxt)[2], f4v (&rb_nxt)[2]) {
    // entering: tile `it` is staged (by every wave, before the barrier below), tile it+1's rows are in flight in *_nxt
    __syncthreads();
    const char* imgA = lds + (it & 1) * (int)W3_LDS_BYTES;
    const char* imgB = imgA + 3 * W3_IMG;
    pl_u4 fa[MI][3], fb[2][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fa[mi][t] = *reinterpret_cast<const pl_u4*>(imgA + t * W3_IMG + w3_off(wm * (16 * MI) + mi * 16 + i16, kk));
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fb[ni][t] = *reinterpret_cast<const pl_u4*>(imgB + t * W3_IMG + w3_off(wn * 32 + ni * 16 + i16, kk));
    if (it + 1 < nk) stage(it + 1, ra_nxt, rb_nxt);   // into the other stage: everyone left it before the barrier above
    if (it + 2 < nk) load(it + 2, ra_cur, rb_cur);     // the set tile `it` came from is free
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][ta], fb[ni][tb], acc[mi][ni]);
    }
  };
  load(0, xa[0], xb[0]);
  load(1, xa[1], xb[1]);
  stage(0, xa[0], xb[0]);
  int it = 0;
  for (; it + 1 < nk; it += 2) {
    step(it, xa[0], xb[0], xa[1], xb[1]);
    step(it + 1, xa[1], xb[1], xa[0], xb[0]);
  }
  if (it < nk) step(it, xa[0], xb[0], xa[1], xb[1]);
  f4v blk[MI * 2]; int rb[MI * 2], cb[MI * 2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + wm * (16 * MI) + mi * 16; cb[mi * 2 + ni] = n0 + wn * 32 + ni * 16;
    }
  store_acc_blocks<MI * 2>(p, C, bz, rb, cb, blk, kk, i16);
}

// The LATENCY-PROOF form for SMALL weight gradients (round 6): the two forms above keep one K-tile of rows in flight in
// registers and rely on 3 - 5 resident workgroups per CU to cover the rest of the memory latency.  The weight gradients of the
// EP step at B = 1024 are 192 - 384 tiles -- one or two workgroups per CU -- and every K-tile then costs a full round trip:
// rocprofv3, stand-alone on a side queue at 1024 x 256 x 768: dWc = dlogits^T z (1.6 GFLOP) 38.8 us, dWv (1.2 GFLOP) 47.4 us,
// i.e. ~1.2 us per K-tile for ~0.3 us of instructions.  Here the raw fp32 K-tiles (32 batch rows x BMT / 64 columns, whole
// 128- / 256-byte row segments) come in by LDS-DMA into a ring of three stages, two K-tiles ahead of their use and at no
// register cost; the waves then read their k-pair x 4 columns from the ring instead of from memory and split / transpose /
// multiply exactly as gemm_tile_b3g does (same plane images, same matrix-instruction order: bit-identical results).
// LDS: 3 x (4 | 8 + 8) KiB ring + 30 KiB images = 66 / 78 KiB.  Both operands T layout, 16-byte aligned, ld % 4 == 0.
constexpr int W3D_NSTG = 3;
template <int BMT> constexpr size_t w3d_lds_bytes() { return (size_t)W3D_NSTG * (32 * BMT * 4 + 32 * 64 * 4) + W3_LDS_BYTES; }
template <int BMT>
__device__ __forceinline__ void gemm_tile_b3d(const GemmParams& p, int bx, int by, int bz, char* lds) {
  constexpr int MI = BMT / 32;
  constexpr int RAW_A = 32 * BMT * 4, RAW_B = 32 * 64 * 4, RAW = RAW_A + RAW_B;
  constexpr int NDA = RAW_A / 1024, ND = NDA + 8, PW = ND / 4;        // 1-KiB DMA pieces per K-tile: 12 / 16, 3 / 4 per wave
  constexpr int CPA = BMT / 4, RPA = 64 / CPA;                        // A: 16-byte chunks per k-row, k-rows per piece
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * BMT, n0 = bx * 64;
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  char* ring = lds;
  char* imgA = lds + W3D_NSTG * RAW;
  char* imgB = imgA + 3 * W3_IMG;
  const int mq = tid & 15, kp = tid >> 4;
  const bool stA = 4 * mq < BMT;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  const int nk = (p.K + 31) / 32;
  const bool one = p.nterms == 1;
  // this wave's DMA pieces: piece i = w + 4 j; i < NDA: k-rows RPA i .. of the A tile, else k-rows 4 (i - NDA) .. of the B tile
  const float* src[PW]; int krow[PW]; int64_t ldp[PW]; int dst[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int i = w + 4 * j;
    if (i < NDA) {
      const int c = m0 + 4 * (lane % CPA);
      krow[j] = RPA * i + lane / CPA; ldp[j] = p.lda; src[j] = A + (c < extA ? c : 0); dst[j] = i * 1024;
    } else {
      const int c = n0 + 4 * (lane & 15);
      krow[j] = 4 * (i - NDA) + (lane >> 4); ldp[j] = p.ldb; src[j] = B + (c < extB ? c : 0); dst[j] = RAW_A + (i - NDA) * 1024;
    }
  }
  auto issue = [&](int t, int slot) {                 // K-tile t (clamped: redundant, never out of range) into ring stage `slot`
    const int tt = t < nk ? t : nk - 1;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      int k = tt * 32 + krow[j];
      k = k < p.K ? k : p.K - 1;
      __builtin_amdgcn_global_load_lds((pl_gptr_t)(src[j] + (int64_t)k * ldp[j]), (pl_lds_ptr_t)(ring + slot * RAW + dst[j]), 16, 0, 0);
    }
  };
  f4v acc[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  issue(1, 1);
  int slot = 0;
  for (int it = 0; it < nk; ++it) {
    pl_dma_wait<PW>();                                // my pieces of tile `it` (tile it+1 may still be in flight)
    // (raw barriers: with LDS-DMA in flight __syncthreads() drains vmcnt(0) -- its workgroup-scope release covers the copies)
    pl_barrier();                                     // everybody's pieces; and every wave has read tile it-1's fragments
    issue(it + 2, slot == 0 ? 2 : slot - 1);          // the stage tile it-1 left (its rows were read before the barrier below, last round)
    const char* raw = ring + slot * RAW;
    f4v xa[2], xb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      xa[h] = *reinterpret_cast<const f4v*>(raw + (2 * kp + h) * (BMT * 4) + (stA ? mq : 0) * 16);
      xb[h] = *reinterpret_cast<const f4v*>(raw + RAW_A + (2 * kp + h) * 256 + mq * 16);
    }
    if (stA) w3_stage_T(imgA, xa, extA, p.K, m0, it * 32, kp, mq, one);
    w3_stage_T(imgB, xb, extB, p.K, n0, it * 32, kp, mq, one);
    pl_barrier();                                     // the plane images are complete; the ring stage is free
    slot = slot == 2 ? 0 : slot + 1;
    pl_u4 fa[MI][3], fb[2][3];
    if (one) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi][0] = *reinterpret_cast<const pl_u4*>(imgA + w3_off(wm * (16 * MI) + mi * 16 + i16, kk));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[ni][0] = *reinterpret_cast<const pl_u4*>(imgB + w3_off(wn * 32 + ni * 16 + i16, kk));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][0], fb[ni][0], acc[mi][ni]);
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fa[mi][t] = *reinterpret_cast<const pl_u4*>(imgA + t * W3_IMG + w3_off(wm * (16 * MI) + mi * 16 + i16, kk));
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fb[ni][t] = *reinterpret_cast<const pl_u4*>(imgB + t * W3_IMG + w3_off(wn * 32 + ni * 16 + i16, kk));
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][ta], fb[ni][tb], acc[mi][ni]);
    }
  }
  pl_dma_wait<0>();                                   // the redundant tail pieces have landed: the ring may be reused
  f4v blk[MI * 2]; int rb[MI * 2], cb[MI * 2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + wm * (16 * MI) + mi * 16; cb[mi * 2 + ni] = n0 + wn * 32 + ni * 16;
    }
  store_acc_blocks<MI * 2>(p, C, bz, rb, cb, blk, kk, i16);
  __syncthreads();                                    // LDS free for the caller's next tile
}

// The WIDE form for the long weight gradients of the matrix-core-bound heads (round 6): 128 x 128 output tile, both operands T
// layout.  dW = dG^T X over the B N = 65536 token rows of the AbMILP / DINOv2-block / DOLG heads (reference poolings/abmilp.py:53-71,
// poolings/dinov2_layers/block.py:86-113 under autograd) re-reads every operand column block once per tile of the other operand:
// on 64 x 64 tiles the 1152 x 1152 x 65536 gradient pulls 10.6 GB through the L2 -> LDS path (16 KiB per K-tile and tile), and the
// five of them in an AbMILP step at 256 x 1152 take 5 x 1.69 ms = 112 TFLOP/s fp32-equivalent (rocprofv3) -- operand-traffic
// bound, not matrix bound (27 % of the bf16 pipe).  A 128 x 128 tile halves the bytes per FLOP and doubles the matrix work per
// barrier pair (96 instead of 24 instructions per wave and K-tile).  Wave (wm, wn) of the 2 x 2 owns 64 x 64 = 4 x 4 blocks;
// plane images of 128 rows (term stride 2 W3_IMG), one LDS stage of 60 KiB, the next K-tile's rows in flight in registers.
constexpr int W3W_TS = 2 * W3_IMG;                                   // bytes per term image (128 rows)
constexpr size_t W3W_LDS_BYTES = 6 * (size_t)W3W_TS;                 // A: h m l | B: h m l = 61440
// A_K / B_K (round 6, late): the K-contiguous layouts too -- the per-image attention products of the AbMILP head (S = q k^T,
// O = A v, dq, dk, dv, dA: 256 x 256 x 1152 / 256 x 1152 x 256 per image) ran on 64 x 64 tiles, L2 -> LDS fill bound.
template <bool A_K = false, bool B_K = false>
__device__ __forceinline__ void gemm_tile_b3w(const GemmParams& p, int bx, int by, int bz, char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * 128, n0 = bx * 128;
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  char* imgA = lds;
  char* imgB = lds + 3 * W3W_TS;
  const int mq = tid & 15, kp = tid >> 4;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  const int nk = (p.K + 31) / 32;
  const bool one = p.nterms == 1;
  f4v acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  f4v xa[2][2], xb[2][2];                            // [64-column half][k of the pair]  (K layout: [64-row half][row of the pair])
  auto load = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (A_K) w3_load_K<64>(A, p.lda, p.M, p.K, m0 + 64 * h, k0, tid, xa[h]);
      else w3_load_T(A, p.lda, extA, p.K, m0 + 64 * h, k0, kp, mq, xa[h]);
      if constexpr (B_K) w3_load_K<64>(B, p.ldb, p.N, p.K, n0 + 64 * h, k0, tid, xb[h]);
      else w3_load_T(B, p.ldb, extB, p.K, n0 + 64 * h, k0, kp, mq, xb[h]);
    }
  };
  load(0);
  for (int it = 0; it < nk; ++it) {
    if (it > 0) __syncthreads();                     // every wave has read tile it-1's fragments
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (A_K) w3_stage_K<64>(imgA + h * W3_IMG, xa[h], p.M, p.K, m0 + 64 * h, it * 32, tid, one, W3W_TS);
      else w3_stage_T(imgA + h * W3_IMG, xa[h], extA, p.K, m0 + 64 * h, it * 32, kp, mq, one, W3W_TS);
      if constexpr (B_K) w3_stage_K<64>(imgB + h * W3_IMG, xb[h], p.N, p.K, n0 + 64 * h, it * 32, tid, one, W3W_TS);
      else w3_stage_T(imgB + h * W3_IMG, xb[h], extB, p.K, n0 + 64 * h, it * 32, kp, mq, one, W3W_TS);
    }
    if (it + 1 < nk) load((it + 1) * 32);            // in flight while this tile is multiplied
    __syncthreads();                                 // the plane images are complete
    if (one) {
      pl_u4 fb1[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb1[ni] = *reinterpret_cast<const pl_u4*>(imgB + w3_off(wn * 64 + ni * 16 + i16, kk));
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const pl_u4 fa1 = *reinterpret_cast<const pl_u4*>(imgA + w3_off(wm * 64 + mi * 16 + i16, kk));
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = pl_mfma(fa1, fb1[ni], acc[mi][ni]);
      }
      continue;
    }
    pl_u4 fb[4][3];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fb[ni][t] = *reinterpret_cast<const pl_u4*>(imgB + t * W3W_TS + w3_off(wn * 64 + ni * 16 + i16, kk));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      pl_u4 fa[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) fa[t] = *reinterpret_cast<const pl_u4*>(imgA + t * W3W_TS + w3_off(wm * 64 + mi * 16 + i16, kk));
      // smallest terms first, as gemm_tile_b3g: lo x hi, hi x lo, mid x mid, mid x hi, hi x mid, hi x hi
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
        const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = pl_mfma(fa[ta], fb[ni][tb], acc[mi][ni]);
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {                   // the epilogue in four groups of four blocks (its loads and stores stay counted)
    f4v blk[4]; int rb[4], cb[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) { blk[ni] = acc[mi][ni]; rb[ni] = m0 + wm * 64 + mi * 16; cb[ni] = n0 + wn * 64 + ni * 16; }
    store_acc_blocks<4>(p, C, bz, rb, cb, blk, kk, i16);
  }
}

// The wide tile for ONE product of the bf16-rounded operands (the AMP-bf16 arithmetic mode, GemmParams.nterms == 1; round 6).
// gemm_tile_b3w with a run-time `one` keeps its 60 KiB of plane images and 248 registers: two workgroups per CU, each K-tile a
// chain of two barriers around 16 matrix instructions per wave (rocprofv3, AbMILP step at 256 x 1152: 0.55 ms per 1152 x 1152 x
// 65536 gradient, 18 % of the matrix pipe).  Here only the rounded operand is staged -- 20 KiB per K-tile -- into TWO image sets,
// so tile it + 1 is written while tile it is multiplied and ONE barrier per K-tile remains; 40 KiB and <= 128 registers let
// three to four workgroups share a CU.  Same rounding, same per-tile instruction order as the `one` branch of gemm_tile_b3w.
constexpr size_t W3W1_LDS_BYTES = 2 * 2 * (size_t)W3W_TS;            // two sets of (A | B) = 40960
__device__ __forceinline__ void w3_stage_T1(char* img, const f4v (&x)[2], int ext, int K, int c0, int k0, int kp, int mq) {
  const bool cok = c0 + 4 * mq < ext;
  const bool k0ok = cok && k0 + 2 * kp < K, k1ok = cok && k0 + 2 * kp + 1 < K;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<unsigned*>(img + w3_off(4 * mq + j, kp >> 2) + 4 * (kp & 3)) = pl_pack_rne(k0ok ? x[0][j] : 0.f, k1ok ? x[1][j] : 0.f);
}
template <bool A_K = false, bool B_K = false>
__device__ __forceinline__ void gemm_tile_b3w1(const GemmParams& p, int bx, int by, int bz, char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * 128, n0 = bx * 128;
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  const int mq = tid & 15, kp = tid >> 4;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  const int nk = (p.K + 31) / 32;
  f4v acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  f4v xa[2][2], xb[2][2];                            // [64-column half][k of the pair]
  auto load = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (A_K) w3_load_K<64>(A, p.lda, p.M, p.K, m0 + 64 * h, k0, tid, xa[h]);
      else w3_load_T(A, p.lda, extA, p.K, m0 + 64 * h, k0, kp, mq, xa[h]);
      if constexpr (B_K) w3_load_K<64>(B, p.ldb, p.N, p.K, n0 + 64 * h, k0, tid, xb[h]);
      else w3_load_T(B, p.ldb, extB, p.K, n0 + 64 * h, k0, kp, mq, xb[h]);
    }
  };
  auto stage = [&](int t) {                          // the rows in registers (K-tile t) -> image set t & 1
    char* imgA = lds + (t & 1) * (2 * W3W_TS);
    char* imgB = imgA + W3W_TS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (A_K) w3_stage_K<64>(imgA + h * W3_IMG, xa[h], p.M, p.K, m0 + 64 * h, t * 32, tid, true);
      else w3_stage_T1(imgA + h * W3_IMG, xa[h], extA, p.K, m0 + 64 * h, t * 32, kp, mq);
      if constexpr (B_K) w3_stage_K<64>(imgB + h * W3_IMG, xb[h], p.N, p.K, n0 + 64 * h, t * 32, tid, true);
      else w3_stage_T1(imgB + h * W3_IMG, xb[h], extB, p.K, n0 + 64 * h, t * 32, kp, mq);
    }
  };
  load(0);
  stage(0);
  if (nk > 1) load(32);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    if (it + 1 < nk) {
      stage(it + 1);                                 // set (it + 1) & 1: last read in iteration it - 1, before the barrier that ended it
      if (it + 2 < nk) load((it + 2) * 32);
    }
    const char* imgA = lds + (it & 1) * (2 * W3W_TS);
    const char* imgB = imgA + W3W_TS;
    pl_u4 fb1[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fb1[ni] = *reinterpret_cast<const pl_u4*>(imgB + w3_off(wn * 64 + ni * 16 + i16, kk));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const pl_u4 fa1 = *reinterpret_cast<const pl_u4*>(imgA + w3_off(wm * 64 + mi * 16 + i16, kk));
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = pl_mfma(fa1, fb1[ni], acc[mi][ni]);
    }
    __syncthreads();                                 // tile it + 1's images are complete; everyone has read tile it's
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    f4v blk[4]; int rb[4], cb[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) { blk[ni] = acc[mi][ni]; rb[ni] = m0 + wm * 64 + mi * 16; cb[ni] = n0 + wn * 64 + ni * 16; }
    store_acc_blocks<4>(p, C, bz, rb, cb, blk, kk, i16);
  }
}

// The PAIRED form for SMALL weight gradients that have a CU to themselves (round 6): 8 waves = two groups of four on ONE 64 x 64
// output tile, alternating K-tiles (the planes kernel's two parities, ep_planes.hip).  gemm_tile_b3g's K-tile is a serial chain --
// raw rows -> split -> transposed LDS write -> barrier -> fragment reads -> 24 matrix instructions -> barrier -- that takes
// ~1.2 us when nothing shares the CU (rocprofv3: dWc 38.8 us, dWv 47.4 us stand-alone at 1024 rows, EXPERIMENTS.md r6.1); here,
// between two workgroup barriers, one group STAGES its next tile (vector ALU + LDS writes) while the other group MULTIPLIES its
// current one (LDS reads + matrix pipe) on the same four SIMDs (wave w and w + 4 share SIMD w % 4).  Raw fp32 K-tiles come in by
// LDS-DMA into a ring of four 16-KiB stages, three tiles ahead; each group owns one set of plane images (30 KiB).  Every output
// element is the sum of the two groups' accumulators: group 1 hands its 16 blocks over through LDS, group 0 adds and stores.
// Same split, same matrix-instruction order per K-tile as gemm_tile_b3g; the sum over K-tiles is taken as (even tiles) + (odd
// tiles) -- a different, fixed order.  T / T layout, 16-byte aligned operands, 64-row tiles.
constexpr int W3P_NSTG = 4;
constexpr int W3P_RAW = 2 * 32 * 64 * 4;                              // one K-tile of both operands, raw fp32: 16384
constexpr size_t W3P_LDS_BYTES = (size_t)W3P_NSTG * W3P_RAW + 2 * W3_LDS_BYTES;    // 65536 + 61440
__device__ __forceinline__ void gemm_tile_b3p(const GemmParams& p, int bx, int by, int bz, char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);             // 0 .. 7
  const int grp = w >> 2, wg4 = w & 3;
  const int wm = wg4 >> 1, wn = wg4 & 1;
  const int m0 = by * 64, n0 = bx * 64;
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  char* ring = lds;
  char* imgA = lds + W3P_NSTG * W3P_RAW + grp * (int)W3_LDS_BYTES;
  char* imgB = imgA + 3 * W3_IMG;
  const int tg = tid & 255;                                           // thread of its group
  const int mq = tg & 15, kp = tg >> 4;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  const int nk = (p.K + 31) / 32;
  const bool one = p.nterms == 1;
  // this wave's two DMA pieces of a K-tile: piece i = w + 8 j; i < 8: k-rows 4 i .. 4 i + 3 of the A tile, else of the B tile
  const float* src[2]; int krow[2]; int64_t ldp[2]; int dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = w + 8 * j;
    if (i < 8) {
      const int c = m0 + 4 * (lane & 15);
      krow[j] = 4 * i + (lane >> 4); ldp[j] = p.lda; src[j] = A + (c < extA ? c : 0); dst[j] = i * 1024;
    } else {
      const int c = n0 + 4 * (lane & 15);
      krow[j] = 4 * (i - 8) + (lane >> 4); ldp[j] = p.ldb; src[j] = B + (c < extB ? c : 0); dst[j] = 8192 + (i - 8) * 1024;
    }
  }
  auto issue = [&](int t) {                           // K-tile t (clamped: redundant, never out of range) into stage t % 4
    const int tt = t < nk ? t : nk - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int k = tt * 32 + krow[j];
      k = k < p.K ? k : p.K - 1;
      __builtin_amdgcn_global_load_lds((pl_gptr_t)(src[j] + (int64_t)k * ldp[j]), (pl_lds_ptr_t)(ring + (t & 3) * W3P_RAW + dst[j]), 16, 0, 0);
    }
  };
  f4v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  issue(0); issue(1); issue(2);
  // interval i (between barriers i and i + 1): group i % 2 stages tile i, the other group multiplies tile i - 1
  for (int i = 0; i <= nk; ++i) {
    pl_dma_wait<4>();                                 // my pieces of tile i (tiles i + 1, i + 2 may be in flight)
    pl_barrier();                                     // tile i landed everywhere; the images staged in interval i - 1 are complete;
                                                      // the fragments read in interval i - 1 are consumed
    if (!(p.ablate & 4)) issue(i + 3);                // into the stage tile i - 1 left (its rows were read in interval i - 1)
    if ((i & 1) == grp) {
      if (i < nk && !(p.ablate & 2)) {
        const char* raw = ring + (i & 3) * W3P_RAW;
        f4v xa[2], xb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          xa[h] = *reinterpret_cast<const f4v*>(raw + (2 * kp + h) * 256 + mq * 16);
          xb[h] = *reinterpret_cast<const f4v*>(raw + 8192 + (2 * kp + h) * 256 + mq * 16);
        }
        w3_stage_T(imgA, xa, extA, p.K, m0, i * 32, kp, mq, one);
        w3_stage_T(imgB, xb, extB, p.K, n0, i * 32, kp, mq, one);
      }
    } else if (i >= 1 && !(p.ablate & 1)) {
      pl_u4 fa[2][3], fb[2][3];
      if (one) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi][0] = *reinterpret_cast<const pl_u4*>(imgA + w3_off(wm * 32 + mi * 16 + i16, kk));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ni][0] = *reinterpret_cast<const pl_u4*>(imgB + w3_off(wn * 32 + ni * 16 + i16, kk));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][0], fb[ni][0], acc[mi][ni]);
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int t = 0; t < 3; ++t) fa[mi][t] = *reinterpret_cast<const pl_u4*>(imgA + t * W3_IMG + w3_off(wm * 32 + mi * 16 + i16, kk));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int t = 0; t < 3; ++t) fb[ni][t] = *reinterpret_cast<const pl_u4*>(imgB + t * W3_IMG + w3_off(wn * 32 + ni * 16 + i16, kk));
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][ta], fb[ni][tb], acc[mi][ni]);
        }
      }
    }
  }
  pl_dma_wait<0>();                                   // the redundant tail pieces have landed: the ring serves the hand-over
  pl_barrier();
  f4v* hand = reinterpret_cast<f4v*>(ring) + (wg4 * 64 + lane) * 4;
  if (grp == 1) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) hand[mi * 2 + ni] = acc[mi][ni];
  }
  pl_barrier();
  if (grp == 1) return;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] += hand[mi * 2 + ni];
  f4v blk[4]; int rb[4], cb[4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + wm * 32 + mi * 16; cb[mi * 2 + ni] = n0 + wn * 32 + ni * 16;
    }
  store_acc_blocks<4>(p, C, bz, rb, cb, blk, kk, i16);
}

// The BARRIER-FREE form (round 6, after the ablation of gemm_tile_b3p -- profiles / EXPERIMENTS.md r6.8: with one workgroup per CU
// the bare barrier interval costs 0.33 us and the phases of the two groups add up instead of overlapping): every WAVE owns a
// 32 x 32 output tile and needs nobody -- no LDS, no barrier.  A k-slow operand is loaded straight into the matrix instruction's
// layout: lane (i16, kk) takes the eight values X[k0 + 8 kk + j][c0 + i16], j = 0 .. 7, as eight dword loads (16 consecutive
// columns x 4 rows = four 64-byte segments per instruction, from L2), splits them in registers (pl_split8) and multiplies.  The next
// K-tile's 32 loads are in flight while this one is split and multiplied.  Every operand block is loaded and split by the two waves
// that share it (twice the vector work of the LDS forms) -- the price of independence; many waves per SIMD hide the rest.
// T / T layout; any alignment of the columns (dword loads); the four waves of a workgroup cover a 64 x 64 tile so that their
// loads share cache lines.
__device__ __forceinline__ void w3f_load_block(const float* __restrict__ X, int64_t ld, int ext, int K, int c0, int k0, int i16, int kk,
                                               float (&v)[8]) {
  const int c = c0 + i16;
  const float* col = X + (c < ext ? c : 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + 8 * kk + j;
    v[j] = col[(int64_t)(k < K ? k : K - 1) * ld];
  }
}
__device__ __forceinline__ void w3f_mask_split(float (&v)[8], bool cok, int K, int k0, int kk, pl_u4 (&t)[3]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (cok && k0 + 8 * kk + j < K) ? v[j] : 0.f;
  pl_split8(v, t);
}
__device__ __forceinline__ void gemm_tile_b3f(const GemmParams& p, int bx, int by, int bz) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m0 = by * 64 + (w >> 1) * 32, n0 = bx * 64 + (w & 1) * 32;
  if (m0 >= p.M || n0 >= p.N) return;                // (wave-uniform: a whole 32 x 32 tile outside the matrix)
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  const int nk = (p.K + 31) / 32;
  const bool one = p.nterms == 1;
  const bool ca0 = m0 + i16 < extA, ca1 = m0 + 16 + i16 < extA, cb0 = n0 + i16 < extB, cb1 = n0 + 16 + i16 < extB;
  f4v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  float xa[2][2][8], xb[2][2][8];                    // [register set][block][j]
  auto load = [&](int t, float (&ra)[2][8], float (&rb)[2][8]) {
    const int k0 = (t < nk ? t : nk - 1) * 32;       // (clamped: redundant, never out of range)
    w3f_load_block(A, p.lda, extA, p.K, m0, k0, i16, kk, ra[0]);
    w3f_load_block(A, p.lda, extA, p.K, m0 + 16, k0, i16, kk, ra[1]);
    w3f_load_block(B, p.ldb, extB, p.K, n0, k0, i16, kk, rb[0]);
    w3f_load_block(B, p.ldb, extB, p.K, n0 + 16, k0, i16, kk, rb[1]);
  };
  auto step = [&](int t, float (&ra)[2][8], float (&rb)[2][8]) {
    pl_u4 fa[2][3], fb[2][3];
    w3f_mask_split(ra[0], ca0, p.K, t * 32, kk, fa[0]);
    w3f_mask_split(ra[1], ca1, p.K, t * 32, kk, fa[1]);
    w3f_mask_split(rb[0], cb0, p.K, t * 32, kk, fb[0]);
    w3f_mask_split(rb[1], cb1, p.K, t * 32, kk, fb[1]);
    if (one) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][0], fb[ni][0], acc[mi][ni]);
      return;
    }
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {                 // smallest terms first, as gemm_tile_b3g
      const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][ta], fb[ni][tb], acc[mi][ni]);
    }
  };
  load(0, xa[0], xb[0]);
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    load(t + 1, xa[1], xb[1]);
    step(t, xa[0], xb[0]);
    load(t + 2, xa[0], xb[0]);
    step(t + 1, xa[1], xb[1]);
  }
  if (t < nk) step(t, xa[0], xb[0]);
  f4v blk[4]; int rb[4], cb[4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + mi * 16; cb[mi * 2 + ni] = n0 + ni * 16; }
  store_acc_blocks<4>(p, C, bz, rb, cb, blk, kk, i16);
}

// ... and with the K range dealt over the four waves of a workgroup (K-tile t to wave t % 4), the workgroup owning ONE 32 x 32
// output tile: a 1024-row gradient of 1000 x 768 is then 768 workgroups = 3072 waves of 8 K-tiles each instead of 1536 waves of
// 32 -- the wave's serial chain (load, split, multiply) is a quarter as long and three waves per SIMD cover each other.  The four
// partial tiles meet in LDS at the end (16 KiB, one barrier), summed in wave order.
__device__ __forceinline__ void gemm_tile_b3fk(const GemmParams& p, int bx, int by, int bz, float* red) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m0 = by * 32, n0 = bx * 32;
  const float* A = p.A + (int64_t)bz * p.sAz;
  const float* B = p.B + (int64_t)bz * p.sBz;
  float* C = p.C + (int64_t)bz * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;
  const int extA = p.extA < p.M ? p.extA : p.M, extB = p.extB < p.N ? p.extB : p.N;
  const int nk = (p.K + 31) / 32;
  const bool one = p.nterms == 1;
  const bool ca0 = m0 + i16 < extA, ca1 = m0 + 16 + i16 < extA, cb0 = n0 + i16 < extB, cb1 = n0 + 16 + i16 < extB;
  f4v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};
  float xa[2][2][8], xb[2][2][8];
  // Row k0 + j of an operand is a UNIFORM base (scalar arithmetic) plus one per-lane 32-bit BYTE offset (8 kk ld + column) that never
  // changes: global_load_dword with an SGPR base.  (The first form computed a 64-bit product per load, 32 per K-tile: ~770 of the
  // ~1600 vector cycles of a wave's K-tile -- rocprofv3: 43 us for the EP step's two gradients at 1024 rows.)  Needs K % 32 == 0
  // and 32-bit offsets (wgrad_pair_ok checks both).
  const bool full = m0 + 32 <= extA && n0 + 32 <= extB;                  // (uniform) no column of the tile is masked
  const int colA = m0 + i16, colB = n0 + i16;
  const unsigned boA0 = 4u * (unsigned)(8 * kk * (int)p.lda + (colA < extA ? colA : 0)), boA1 = 4u * (unsigned)(8 * kk * (int)p.lda + (colA + 16 < extA ? colA + 16 : 0));
  const unsigned boB0 = 4u * (unsigned)(8 * kk * (int)p.ldb + (colB < extB ? colB : 0)), boB1 = 4u * (unsigned)(8 * kk * (int)p.ldb + (colB + 16 < extB ? colB + 16 : 0));
  auto ldf = [](const float* base, unsigned bo) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + bo); };
  auto load = [&](int t, float (&ra)[2][8], float (&rb)[2][8]) {
    const int k0 = (t < nk ? t : nk - 1) * 32;
    const float* Ak = A + (int64_t)k0 * p.lda;
    const float* Bk = B + (int64_t)k0 * p.ldb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* ar = Ak + (int64_t)j * p.lda;       // uniform
      const float* br = Bk + (int64_t)j * p.ldb;
      ra[0][j] = ldf(ar, boA0); ra[1][j] = ldf(ar, boA1);
      rb[0][j] = ldf(br, boB0); rb[1][j] = ldf(br, boB1);
    }
  };
  auto step = [&](int t, float (&ra)[2][8], float (&rb)[2][8]) {
    pl_u4 fa[2][3], fb[2][3];
    if (!full) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ra[0][j] = ca0 ? ra[0][j] : 0.f; ra[1][j] = ca1 ? ra[1][j] : 0.f;
        rb[0][j] = cb0 ? rb[0][j] : 0.f; rb[1][j] = cb1 ? rb[1][j] : 0.f;
      }
    }
    pl_split8(ra[0], fa[0]); pl_split8(ra[1], fa[1]); pl_split8(rb[0], fb[0]); pl_split8(rb[1], fb[1]);
    if (one) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][0], fb[ni][0], acc[mi][ni]);
      return;
    }
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = pl_mfma(fa[mi][ta], fb[ni][tb], acc[mi][ni]);
    }
  };
  int t = w;                                          // this wave's K-tiles: w, w + 4, ...
  if (t < nk) load(t, xa[0], xb[0]);
  for (; t + 4 < nk; t += 8) {
    load(t + 4, xa[1], xb[1]);
    step(t, xa[0], xb[0]);
    if (t + 8 < nk) load(t + 8, xa[0], xb[0]);
    step(t + 4, xa[1], xb[1]);
  }
  if (t < nk) step(t, xa[0], xb[0]);
  // the four partial tiles: waves 1 .. 3 hand theirs over, wave 0 sums them in wave order and stores
  f4v* mine = reinterpret_cast<f4v*>(red) + ((w - 1) * 64 + lane) * 4;
  if (w > 0) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) mine[mi * 2 + ni] = acc[mi][ni];
  }
  __syncthreads();
  if (w > 0) return;
#pragma unroll
  for (int ww = 0; ww < 3; ++ww) {
    const f4v* o = reinterpret_cast<const f4v*>(red) + (ww * 64 + lane) * 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] += o[mi * 2 + ni];
  }
  f4v blk[4]; int rb[4], cb[4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + mi * 16; cb[mi * 2 + ni] = n0 + ni * 16; }
  store_acc_blocks<4>(p, C, bz, rb, cb, blk, kk, i16);
}

// the weight-gradient form (both operands T layout): what the token passes run as side work
template <int BMT>
__device__ __forceinline__ void gemm_tile_b3(const GemmParams& p, int bx, int by, int bz, char* lds) {
  gemm_tile_b3g<false, false, BMT>(p, bx, by, bz, lds);
}

}  // namespace ep
