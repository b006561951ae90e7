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
      const float bv = p.bias ? p.bias[(int64_t)z * p.sBiasz + col] : 0.f;
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

// vector loads need 16-byte aligned rows on every operand
static bool vec_ok(const float* ptr, int64_t ld, int64_t sz, int inner) {
  return aligned16(ptr) && ld % 4 == 0 && sz % 4 == 0 && inner % 4 == 0;
}

// can this contraction run as a side task of the second token pass (ep_side.h: T/T layout, vector loads)?
bool gemm_side_ok(const GemmParams& p, bool a_k, bool b_k) {
  return !a_k && !b_k && p.M > 0 && p.N > 0 && p.K > 0 && !p.bias &&
         vec_ok(p.A, p.lda, p.sAz, p.extA) && vec_ok(p.B, p.ldb, p.sBz, p.extB);
}

int gemm(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
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
