// Device-side building blocks shared by the stand-alone kernels (ep_gemm.hip, ep_tail.hip) and by the
// second token pass, which runs the weight-gradient contractions as extra workgroups of its own launch
// (ep_pool_stream.hip): the exact-fp32 matrix-core tile, the column sum and the statistics fold.
#pragma once
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDK = BK + 2;    // K-layout row stride (floats)
constexpr int LDT = BM + 16;   // T-layout row stride (floats)


// ---- epilogue of a 16x16x4-MFMA tile ---------------------------------------------------------
// Block b of the NB accumulator blocks of a wave covers rows rowb[b] + 4 kk + r (r = 0..3) and column colb[b] + i16.
// Everything that LOADS (bias, the old values when accumulating) is issued first and waited for once; then the stores go
// out back to back.  With a conditional load in front of every store (`if (accumulate) v += *c; *c = v;`) the compiler
// has to wait vmcnt(0) at each join -- vmcnt counts loads and stores in order, so every store waited for the round trip
// of the one before it: 16 serialised write latencies at the end of every tile.
template <int NB>
__device__ __forceinline__ void store_acc_blocks(const GemmParams& p, float* __restrict__ C, int z, const int (&rowb)[NB],
                                                 const int (&colb)[NB], const f4v (&acc)[NB], int kk, int i16) {
  // loads are unconditional (addresses clamped into the matrix) and every value is finished in straight-line code, so
  // the one wait for them sits in front of the first store and the guarded stores themselves wait for nothing
  float v[NB][4];
  bool cok[NB];
  int64_t off[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = colb[b] + i16;
    cok[b] = col < p.N;
    const int colc = cok[b] ? col : p.N - 1;
    const float bv = p.bias ? p.bias[(int64_t)z * p.sBiasz + colc] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rowb[b] + kk * 4 + r;
      off[b][r] = (int64_t)(row < p.M ? row : p.M - 1) * p.ldc + colc;
      v[b][r] = p.alpha * acc[b][r] + bv;
    }
    if (p.bn_rm) {                                     // eval-mode BatchNorm of the output column (GemmParams.bn_rm)
      const float rm = p.bn_rm[(int64_t)z * p.sBiasz + colc], sd = sqrtf(p.bn_rv[(int64_t)z * p.sBiasz + colc] + p.bn_eps);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[b][r] = (v[b][r] - rm) / sd;
    }
  }
  if (p.accumulate) {
    float old[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) old[b][r] = C[off[b][r]];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[b][r] += old[b][r];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (cok[b] && rowb[b] + kk * 4 + r < p.M) C[off[b][r]] = v[b][r];
}

// ---- global -> register tile loads ---------------------------------------------------------
// K layout: 64 rows x 32 k ; thread handles float4 (row = idx/8, kq = idx%8), idx = tid + 256 r
template <bool VEC, int RT>
__device__ __forceinline__ void load_K(const float* __restrict__ base, int64_t ld, int rows, int K,
                                       int r0, int k0, int tid, f4v (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < RT / 32; ++r) {
    const int idx = tid + 256 * r;
    const int row = r0 + (idx >> 3), k = k0 + 4 * (idx & 7);
    f4v v = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
      // branch-free: out-of-range lanes load a valid address and are zeroed afterwards, so the
      // prefetch loads stay straight-line code and the compiler can count them exactly (vmcnt(N))
      const int rowc = row < rows ? row : rows - 1;
      v = *reinterpret_cast<const f4v*>(base + (int64_t)rowc * ld + (k < K ? k : 0));   // zeroed in store_K
    } else if (row < rows) {
      const float* ptr = base + (int64_t)row * ld + k;
      if (k + 0 < K) v.x = ptr[0];
      if (k + 1 < K) v.y = ptr[1];
      if (k + 2 < K) v.z = ptr[2];
      if (k + 3 < K) v.w = ptr[3];
    }
    reg[r] = v;
  }
}
// T layout: 32 k-rows x 64 r ; thread handles float4 (krow = idx/16, rq = idx%16)
template <bool VEC, int RT>
__device__ __forceinline__ void load_T(const float* __restrict__ base, int64_t ld, int ext, int K,
                                       int r0, int k0, int tid, f4v (&reg)[2]) {
  constexpr int QPR = RT / 4;             // float4 per k-row (16 or 8)
#pragma unroll
  for (int r = 0; r < RT / 32; ++r) {
    const int idx = tid + 256 * r;
    const int k = k0 + idx / QPR, rr = r0 + 4 * (idx % QPR);
    f4v v = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
      const int kc = k < K ? k : K - 1;
      v = *reinterpret_cast<const f4v*>(base + (int64_t)kc * ld + (rr < ext ? rr : 0));  // zeroed in store_T
    } else if (k < K) {
      const float* ptr = base + (int64_t)k * ld + rr;
      if (rr + 0 < ext) v.x = ptr[0];
      if (rr + 1 < ext) v.y = ptr[1];
      if (rr + 2 < ext) v.z = ptr[2];
      if (rr + 3 < ext) v.w = ptr[3];
    }
    reg[r] = v;
  }
}
// the out-of-range mask is applied here, at the point the loaded registers are consumed anyway
// (a select right after the load would force the load to complete immediately)
template <int RT>
__device__ __forceinline__ void store_K(float* lds, int tid, const f4v (&reg)[2], int rows, int K, int r0, int k0) {
#pragma unroll
  for (int r = 0; r < RT / 32; ++r) {
    const int idx = tid + 256 * r;
    const bool ok = (r0 + (idx >> 3)) < rows && (k0 + 4 * (idx & 7)) < K;
    const f4v v = ok ? reg[r] : f4v{0.f, 0.f, 0.f, 0.f};
    float* d = lds + (idx >> 3) * LDK + 4 * (idx & 7);
    *reinterpret_cast<f2*>(d) = f2{v.x, v.y};
    *reinterpret_cast<f2*>(d + 2) = f2{v.z, v.w};
  }
}
template <int RT>
__device__ __forceinline__ void store_T(float* lds, int tid, const f4v (&reg)[2], int ext, int K, int r0, int k0) {
  constexpr int QPR = RT / 4;
#pragma unroll
  for (int r = 0; r < RT / 32; ++r) {
    const int idx = tid + 256 * r;
    const bool ok = (k0 + idx / QPR) < K && (r0 + 4 * (idx % QPR)) < ext;
    const f4v v = ok ? reg[r] : f4v{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f4v*>(lds + (idx / QPR) * LDT + 4 * (idx % QPR)) = v;
  }
}

constexpr int LDS_OPERAND = (64 * LDK > 32 * LDT) ? 64 * LDK : 32 * LDT;   // floats per operand image

// BMT = rows of the output tile (64, or 32 when a 64-row tiling would leave CUs with a single
// workgroup: two resident workgroups per CU hide each other's LDS / barrier latency)
template <bool A_K, bool B_K, bool VEC, int BMT>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, int bx, int by, int bz,
                                          float (*lds)[2][LDS_OPERAND]) {   // lds[buffer][A|B][...]
  constexpr int MI = BMT / 32;            // 16-row MFMA blocks per wave along M
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * BMT, n0 = bx * BN;
  const int z = bz;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;

  f4v acc[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};

  // Register ring of depth 3: the global loads of K-tile it+3 are issued while tile `it` is
  // multiplied, so a load has ~2-3 iterations (2-3k cycles of MFMA) to land before it is copied
  // into LDS one iteration ahead of its use.  R[j] indices are static (loop unrolled by 3).
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
  auto compute = [&](int buf) {
    const float* As = lds[buf][0];
    const float* Bs = lds[buf][1];
    // ALL fragments of the K-tile are read up front (8 k-steps x (MI + 2) values per lane), the matrix instructions
    // follow with counted waits: read just in front of their use (2 reads, wait, 4 MFMAs, 2 reads, wait, ...) the matrix
    // pipe sat idle for an LDS round trip in front of every group of four -- the weight-gradient side work of the
    // second token pass took 65 us on an otherwise EMPTY chip for 27 us of matrix time.
    float af[BK / 4][MI], bf[BK / 4][2];
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * (16 * MI) + mi * 16 + i16;
        af[s][mi] = A_K ? As[row * LDK + 4 * s + kk] : As[(4 * s + kk) * LDT + row];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = wn * 32 + ni * 16 + i16;
        bf[s][ni] = B_K ? Bs[col * LDK + 4 * s + kk] : Bs[(4 * s + kk) * LDT + col];
      }
    }
    __builtin_amdgcn_sched_barrier(0);               // (the scheduler would sink the reads back in front of their uses)
#pragma unroll
    for (int s = 0; s < BK / 4; ++s)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][mi], bf[s][ni], acc[mi][ni], 0, 0, 0);
  };

  const int nk = (p.K + BK - 1) / BK;
  auto tile_k0 = [&](int it) { return (it < nk ? it : nk - 1) * BK; };   // clamped: redundant, never out of range
  gload(0, ra[0], rb[0]);
  lstore(0, ra[0], rb[0], 0);
  gload(tile_k0(1), ra[1], rb[1]);
  gload(tile_k0(2), ra[2], rb[2]);
  __syncthreads();
  // One step = prefetch tile it+3 (unconditionally: straight-line code lets the compiler wait with an
  // exact vmcnt(N) instead of draining), multiply tile it, stage tile it+1 into the other LDS buffer.
#define EP_GEMM_STEP(IT, J)                                                 \
  {                                                                         \
    gload(tile_k0((IT) + 3), ra[J], rb[J]);                                 \
    compute((IT) & 1);                                                      \
    lstore(((IT) + 1) & 1, ra[((J) + 1) % 3], rb[((J) + 1) % 3], ((IT) + 1) * BK); \
    __syncthreads();                                                        \
  }
  int it = 0;
  for (; it + 2 < nk; it += 3) {
    EP_GEMM_STEP(it, 0)
    EP_GEMM_STEP(it + 1, 1)
    EP_GEMM_STEP(it + 2, 2)
  }
  if (it < nk) {
    EP_GEMM_STEP(it, 0)
    if (it + 1 < nk) EP_GEMM_STEP(it + 1, 1)
  }
#undef EP_GEMM_STEP
  // epilogue: D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + r
  {
    f4v blk[MI * 2]; int rb[MI * 2], cb[MI * 2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        blk[mi * 2 + ni] = acc[mi][ni]; rb[mi * 2 + ni] = m0 + wm * (16 * MI) + mi * 16; cb[mi * 2 + ni] = n0 + wn * 32 + ni * 16;
      }
    store_acc_blocks<MI * 2>(p, C, z, rb, cb, blk, kk, i16);
  }
}


// ---- column sum / statistics fold (256 threads) -----------------------------------------------
constexpr int CG = 16;   // columns per workgroup
constexpr int RL = 16;   // row lanes per workgroup  (CG*RL = 256 threads)

// sum over the RL row-lanes of a workgroup for each of its CG columns; result valid for ty == 0
__device__ __forceinline__ float colreduce(float v, float (*sm)[CG], int tx, int ty) {
  sm[ty][tx] = v;
  __syncthreads();
  float s = 0.f;
  if (ty == 0) {
#pragma unroll
    for (int r = 0; r < RL; ++r) s += sm[r][tx];
  }
  __syncthreads();
  return s;
}

// out[col] (+)= sum_b src[b*ld + col] for the CG columns of block bx
__device__ __forceinline__ void colsum_block(const float* __restrict__ src, int B, int ncol, int ld, int accumulate,
                                             float* __restrict__ out, int bx, float (*sm)[CG]) {
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int col = bx * CG + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < ncol) {
    int b = ty;
    for (; b + 3 * RL < B; b += 4 * RL) {
      s0 += src[(int64_t)b * ld + col];
      s1 += src[(int64_t)(b + RL) * ld + col];
      s2 += src[(int64_t)(b + 2 * RL) * ld + col];
      s3 += src[(int64_t)(b + 3 * RL) * ld + col];
    }
    for (; b < B; b += RL) s0 += src[(int64_t)b * ld + col];
  }
  const float s = colreduce((s0 + s1) + (s2 + s3), sm, tx, ty);
  if (ty == 0 && col < ncol) out[col] = accumulate ? out[col] + s : s;
}

// stats[0..3] += sum_b rowstat[b][0..3]   (one workgroup, fixed order: reproducible)
__device__ __forceinline__ void ce_stats_block(const float* __restrict__ rowstat, int B, float* __restrict__ stats,
                                               f4* sm) {
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += 256) s += *reinterpret_cast<const f4*>(rowstat + (int64_t)b * 4);
  s.x = wave_sum(s.x); s.y = wave_sum(s.y); s.z = wave_sum(s.z); s.w = wave_sum(s.w);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const f4 t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    stats[0] += t.x; stats[1] += t.y; stats[2] += t.z; stats[3] += t.w;
  }
}

}  // namespace ep
