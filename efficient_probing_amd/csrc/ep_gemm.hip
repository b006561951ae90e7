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
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDK = BK + 2;    // K-layout row stride (floats)
constexpr int LDT = BM + 16;   // T-layout row stride (floats)


// ---- global -> register tile loads ---------------------------------------------------------
// K layout: 64 rows x 32 k ; thread handles float4 (row = idx/8, kq = idx%8), idx = tid + 256 r
template <bool VEC>
__device__ __forceinline__ void load_K(const float* __restrict__ base, int64_t ld, int rows, int K,
                                       int r0, int k0, int tid, f4v (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = tid + 256 * r;
    const int row = r0 + (idx >> 3), k = k0 + 4 * (idx & 7);
    f4v v = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const float* ptr = base + (int64_t)row * ld + k;
      if (VEC) {
        if (k < K) v = *reinterpret_cast<const f4v*>(ptr);
      } else {
        if (k + 0 < K) v.x = ptr[0];
        if (k + 1 < K) v.y = ptr[1];
        if (k + 2 < K) v.z = ptr[2];
        if (k + 3 < K) v.w = ptr[3];
      }
    }
    reg[r] = v;
  }
}
// T layout: 32 k-rows x 64 r ; thread handles float4 (krow = idx/16, rq = idx%16)
template <bool VEC>
__device__ __forceinline__ void load_T(const float* __restrict__ base, int64_t ld, int ext, int K,
                                       int r0, int k0, int tid, f4v (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = tid + 256 * r;
    const int k = k0 + (idx >> 4), rr = r0 + 4 * (idx & 15);
    f4v v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      const float* ptr = base + (int64_t)k * ld + rr;
      if (VEC) {
        if (rr < ext) v = *reinterpret_cast<const f4v*>(ptr);
      } else {
        if (rr + 0 < ext) v.x = ptr[0];
        if (rr + 1 < ext) v.y = ptr[1];
        if (rr + 2 < ext) v.z = ptr[2];
        if (rr + 3 < ext) v.w = ptr[3];
      }
    }
    reg[r] = v;
  }
}
__device__ __forceinline__ void store_K(float* lds, int tid, const f4v (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = tid + 256 * r;
    float* d = lds + (idx >> 3) * LDK + 4 * (idx & 7);
    *reinterpret_cast<f2*>(d) = f2{reg[r].x, reg[r].y};
    *reinterpret_cast<f2*>(d + 2) = f2{reg[r].z, reg[r].w};
  }
}
__device__ __forceinline__ void store_T(float* lds, int tid, const f4v (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = tid + 256 * r;
    *reinterpret_cast<f4v*>(lds + (idx >> 4) * LDT + 4 * (idx & 15)) = reg[r];
  }
}

constexpr int LDS_OPERAND = (64 * LDK > 32 * LDT) ? 64 * LDK : 32 * LDT;   // floats per operand image

template <bool A_K, bool B_K, bool VEC>
__global__ __launch_bounds__(256) void ep_gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][LDS_OPERAND];   // [buffer][A|B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
  const float* A = p.A + (int64_t)z * p.sAz;
  const float* B = p.B + (int64_t)z * p.sBz;
  float* C = p.C + (int64_t)z * p.sCz;
  const int i16 = lane & 15, kk = lane >> 4;

  f4v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};

  // Register ring of depth 3: the global loads of K-tile it+3 are issued while tile `it` is
  // multiplied, so a load has ~2-3 iterations (2-3k cycles of MFMA) to land before it is copied
  // into LDS one iteration ahead of its use.  R[j] indices are static (loop unrolled by 3).
  f4v ra[3][2], rb[3][2];
  auto gload = [&](int k0, f4v (&xa)[2], f4v (&xb)[2]) {
    if (A_K) load_K<VEC>(A, p.lda, p.M, p.K, m0, k0, tid, xa);
    else load_T<VEC>(A, p.lda, p.extA, p.K, m0, k0, tid, xa);
    if (B_K) load_K<VEC>(B, p.ldb, p.N, p.K, n0, k0, tid, xb);
    else load_T<VEC>(B, p.ldb, p.extB, p.K, n0, k0, tid, xb);
  };
  auto lstore = [&](int buf, const f4v (&xa)[2], const f4v (&xb)[2]) {
    if (A_K) store_K(lds[buf][0], tid, xa); else store_T(lds[buf][0], tid, xa);
    if (B_K) store_K(lds[buf][1], tid, xb); else store_T(lds[buf][1], tid, xb);
  };
  auto compute = [&](int buf) {
    const float* As = lds[buf][0];
    const float* Bs = lds[buf][1];
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      float af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = wm * 32 + mi * 16 + i16;
        af[mi] = A_K ? As[row * LDK + 4 * s + kk] : As[(4 * s + kk) * LDT + row];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = wn * 32 + ni * 16 + i16;
        bf[ni] = B_K ? Bs[col * LDK + 4 * s + kk] : Bs[(4 * s + kk) * LDT + col];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  gload(0, ra[0], rb[0]);
  lstore(0, ra[0], rb[0]);
  if (nk > 1) gload(BK, ra[1], rb[1]);
  if (nk > 2) gload(2 * BK, ra[2], rb[2]);
  __syncthreads();
#define EP_GEMM_STEP(IT, J)                                                   \
  if ((IT) < nk) {                                                            \
    if ((IT) + 3 < nk) gload(((IT) + 3) * BK, ra[J], rb[J]);                  \
    compute((IT) & 1);                                                        \
    if ((IT) + 1 < nk) lstore(((IT) + 1) & 1, ra[((J) + 1) % 3], rb[((J) + 1) % 3]); \
    __syncthreads();                                                          \
  }
  for (int it = 0; it < nk; it += 3) {
    EP_GEMM_STEP(it, 0)
    EP_GEMM_STEP(it + 1, 1)
    EP_GEMM_STEP(it + 2, 2)
  }
#undef EP_GEMM_STEP
  // epilogue: D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 32 + ni * 16 + i16;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + mi * 16 + kk * 4 + r;
        if (row < p.M) {
          float* c = C + (int64_t)row * p.ldc + col;
          float v = p.alpha * acc[mi][ni][r] + bv;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
    }
}

int gemm(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, batch);
  // vector loads need 16-byte aligned rows on every operand
  auto vec_ok = [](const float* ptr, int64_t ld, int64_t sz, int inner) {
    return aligned16(ptr) && ld % 4 == 0 && sz % 4 == 0 && inner % 4 == 0;
  };
  const bool vec = vec_ok(p.A, p.lda, p.sAz, a_k ? p.K : p.extA) && vec_ok(p.B, p.ldb, p.sBz, b_k ? p.K : p.extB);
  size_t pad = 0;
  if (const char* e = getenv("EP_GEMM_PADLDS")) pad = (size_t)atoi(e) * 1024;
  if (pad) {
#define EP_GEMM_ATTR(AK, BK_, V) hipFuncSetAttribute((const void*)ep_gemm_kernel<AK, BK_, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad)
    EP_GEMM_ATTR(true, true, true); EP_GEMM_ATTR(true, false, true); EP_GEMM_ATTR(false, true, true); EP_GEMM_ATTR(false, false, true);
#undef EP_GEMM_ATTR
  }
#define EP_GEMM_LAUNCH(AK, BK_, V) hipLaunchKernelGGL((ep_gemm_kernel<AK, BK_, V>), grid, dim3(256), pad, st, p)
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
  EP_LAUNCH_CHECK("ep_gemm_kernel");
  return 0;
}

}  // namespace ep
