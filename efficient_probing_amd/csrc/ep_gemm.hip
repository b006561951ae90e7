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
__global__ __launch_bounds__(256) void ep_gemm_kernel(GemmParams p) {
  constexpr int MI = BMT / 32;            // 16-row MFMA blocks per wave along M
  __shared__ __attribute__((aligned(16))) float lds[2][2][LDS_OPERAND];   // [buffer][A|B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * BMT, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
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
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      float af[MI], bf[2];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * (16 * MI) + mi * 16 + i16;
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
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 32 + ni * 16 + i16;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (16 * MI) + mi * 16 + kk * 4 + r;
        if (row < p.M) {
          float* c = C + (int64_t)row * p.ldc + col;
          float v = p.alpha * acc[mi][ni][r] + bv;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant (8 waves): waves 4-7 only stage operands (global -> registers, three tiles
// ahead -> LDS, one tile ahead), waves 0-3 only read LDS and issue MFMAs.  The staging work of the
// next tile runs on the same SIMDs in the shadow of the current tile's MFMAs, and the matrix waves
// never wait on global memory: one s_barrier per K-tile is their only synchronisation.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
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

int gemm(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  // vector loads need 16-byte aligned rows on every operand
  auto vec_ok = [](const float* ptr, int64_t ld, int64_t sz, int inner) {
    return aligned16(ptr) && ld % 4 == 0 && sz % 4 == 0 && inner % 4 == 0;
  };
  const bool vec = vec_ok(p.A, p.lda, p.sAz, a_k ? p.K : p.extA) && vec_ok(p.B, p.ldb, p.sBz, b_k ? p.K : p.extB);
  const long tiles64 = (long)((p.N + BN - 1) / BN) * ((p.M + 63) / 64) * batch;
  static int force_bm = -1;
  if (force_bm < 0) { const char* e = getenv("EP_GEMM_BM"); force_bm = e ? atoi(e) : 0; }
  static int use_ws = -1;
  if (use_ws < 0) { const char* e = getenv("EP_GEMM_WS"); use_ws = e ? atoi(e) : 1; }
  const bool small = force_bm ? (force_bm == 32) : (tiles64 < 2L * cu_count());
  // wave-specialised kernel: pays off for long K on the critical path; bit 1 of EP_GEMM_WS also enables it
  // for the weight-gradient contractions that run beside the second token pass
  const bool ws_ok = use_ws && vec && !force_bm && p.K >= 256 && (!p.side || (use_ws & 2));
  if (ws_ok) gemm_launch_ws(a_k, b_k, p, batch, st);
  else if (small) gemm_launch<32>(a_k, b_k, vec, p, batch, st);
  else gemm_launch<64>(a_k, b_k, vec, p, batch, st);
  EP_LAUNCH_CHECK("ep_gemm_kernel");
  return 0;
}

}  // namespace ep
