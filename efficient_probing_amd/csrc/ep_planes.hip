// EXPERIMENT (opt-in: EP_GEMM_PLANES=1; the default train step uses the f32 contraction kernel of ep_gemm.hip).
// fp32 contractions against PRE-SPLIT weights on the gfx950 BF16 matrix cores, at fp32 accuracy.
//
// The head's four critical-path contractions (value projection y = P Wv_q^T, logits = z Wc^T, dz = dlogits Wc,
// dP = dy Wv_q -- reference poolings/ep.py:40, probe_heads.py:76 and their autograd) all multiply an ACTIVATION matrix
// (1024 rows, K contiguous) with a WEIGHT matrix.  On the f32 matrix instruction (v_mfma_f32_16x16x4_f32, the fp32
// vector rate) they take 5.3 us + 0.64 us per 64x64x32 K-tile, about 20 us each.  The bf16 instruction is 16x faster per
// FLOP, and every fp32 value is the EXACT sum of three bf16 values (8 + 8 + 8 significant bits, round-to-nearest terms):
//   x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = x - h - m                  (both subtractions are exact)
// A bf16 x bf16 product is exact in fp32, so
//   a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [am*bl + al*bm + al*bl]
// where the bracket is <= 2^-24 |a*b| and of either sign (dropped): six v_mfma_f32_16x16x32_bf16 per 16x16x32 block
// replace eight f32 instructions at 96 instead of 256 matrix cycles, with fp32 accumulation
// (tests/test_gpu_planes.py: error against float64 at or below the f32 kernel's).
//
// What was tried and measured on the way (MI355X, rocprofv3 device durations of the 1024 x 1000 x 768 logits contraction;
// f32 kernel 20.4 us; operand ring alone, no arithmetic, 10.5 us):
//   1. both operands split in registers by every wave that multiplies them: 19.5 us -- bound by the split's vector
//      instructions (132 per wave and K-tile);
//   2. both operands split once per workgroup, the terms handed over through LDS plane images: 20.0 us (21.5 us with
//      32-row tiles, two workgroups per CU) -- the round trip adds 6 bytes per element of LDS traffic;
//   3. THIS file: the weights arrive split (planes in global memory, written once per step), the activations are split in
//      registers (one 16-row block per wave, 36 vector instructions per K-tile): 21.1 us with the multiply waves also
//      issuing the DMA, 17.1 us with four loader waves and double-buffered fragments (dz: 20.3 against 25.3 us).
// Every variant sits near 0.5 .. 0.64 us per K-tile although none of matrix pipe (19 % busy), LDS (27 %) and vector issue
// is saturated: with ONE workgroup per CU the K-tile step is a chain of latencies (barrier, LDS reads, split, 6-deep MFMA
// chains) at the ~1.65 GHz the chip holds under this load.  In the whole train step the planes path is slower than the f32
// kernels (0.490 against 0.467 ms per step at 1024 x 256 x 768): the split launch (10 us, hidden only partly beside the
// first token pass), the half-empty 96-column tiles of the per-query projection and the 3-K-tile dP contraction (1536
// workgroups of 100 KiB LDS, one per CU) eat what logits and dz gain.  Kept as an opt-in with its tests; what remains to
// try is a deeper software pipeline inside the multiply waves (split of tile t+1 under the MFMAs of tile t).
//
// The contraction kernel
//   * streams the fp32 activation tile AND the weight planes into LDS by LDS-DMA (20 KiB per K-tile), issued by four
//     loader waves (an LDS-DMA instruction costs its issuing wave ~100 cycles);
//   * splits only its own activation fragment in registers;
//   * reads the weight operands ready-made (linear ds_read_b128, no vector work, no LDS write-back);
//   * issues 6 v_mfma_f32_16x16x32_bf16 per 16x16x32 block pair.
//
// Plane format (ep_planes_split_kernel): for a row-major matrix W (R x K) the three terms hi / mid / lo of every element
// as bf16, [term][row][Kp] with Kp = K rounded up to 32 (zero padded), the k-order PERMUTED inside every group of 32:
// position 8 kk + 4 g + j holds k = 16 g + 4 kk + j.  That is the k-assignment the activation fragment reads deliver
// (lane group kk of the MFMA reads chunk 4 g + kk of the swizzled fp32 image), so lane (i, kk) of a block finds its
// eight weight values as 16 contiguous bytes -- one DMA lane, one ds_read_b128 lane.  The transposed orientation
// (planes of W^T, for the two backward contractions that sum over W's row index) is produced in the same launch through
// an LDS transpose.
#include "ep_side.h"

namespace ep {

typedef __attribute__((address_space(3))) void* pl_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pl_gptr_t;
typedef __bf16 pl_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 pl_bf2 __attribute__((ext_vector_type(2)));
typedef unsigned pl_u4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void pl_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// two fp32 values -> one register holding their bf16 roundings (element 0 in the low half): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pl_pack_rne(float v0, float v1) {
  typedef float pl_f2 __attribute__((ext_vector_type(2)));
  const pl_f2 v = {v0, v1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pl_bf2));
}
// two fp32 values -> their three bf16 terms, packed: x = h + m + l exactly (round-to-nearest terms)
__device__ __forceinline__ void pl_split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  h = pl_pack_rne(v0, v1);
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);          // exact
  m = pl_pack_rne(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);          // exact, <= 8 bits
  l = pl_pack_rne(s0, s1);
}
// eight fp32 values in MFMA element order (e = 4 g + j) -> three bf16x8 operands
__device__ __forceinline__ void pl_split8(const float (&v)[8], pl_u4 (&t)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h, m, l;
    pl_split2(v[2 * q], v[2 * q + 1], h, m, l);
    t[0][q] = h; t[1][q] = m; t[2][q] = l;
  }
}
__device__ __forceinline__ f4v pl_mfma(pl_u4 a, pl_u4 b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pl_bf8, a), __builtin_bit_cast(pl_bf8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// weights -> planes, both orientations.  One 64 x 64 tile of W per workgroup (256 threads), blockIdx.z = matrix.
// ---------------------------------------------------------------------------------------------------------------------
struct PlaneJob {
  const float* W; int R, K; int64_t ldw;            // row-major R x K
  uint16_t* pn; int64_t pn_term, pn_ld;             // planes of W   : [3][R][pn_ld],  pn_ld = round_up(K, 32)
  uint16_t* pt; int64_t pt_term, pt_ld;             // planes of W^T : [3][K][pt_ld],  pt_ld = round_up(R, 32)
};
struct PlaneJobs { PlaneJob j[4]; int n; };

__global__ __launch_bounds__(256) void ep_planes_split_kernel(PlaneJobs jobs) {
  __shared__ float tile[64][65];
  const PlaneJob& jb = jobs.j[blockIdx.z];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (r0 >= jb.R || c0 >= jb.K) return;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) {                     // 64 x 64 floats, 16 per thread, coalesced along the row
    const int e = tid + 256 * i, r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < jb.R && c0 + c < jb.K) ? jb.W[(int64_t)(r0 + r) * jb.ldw + c0 + c] : 0.f;
  }
  __syncthreads();
  // 512 work items per orientation: (line, group of 32 along the contraction index, lane group kk) -> 8 values -> 3 x 16 bytes
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, line = e >> 3, grp = (e >> 2) & 1, kk = e & 3;
    float v[8];
    pl_u4 t[3];
    // natural orientation: line = row of W, contraction index = column
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * g + j] = tile[line][32 * grp + 16 * g + 4 * kk + j];
    pl_split8(v, t);
    if (r0 + line < jb.R && c0 + 32 * grp < jb.pn_ld && jb.pn) {          // (the zero padding up to pn_ld is written too)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
        *reinterpret_cast<pl_u4*>(jb.pn + tm * jb.pn_term + (int64_t)(r0 + line) * jb.pn_ld + c0 + 32 * grp + 8 * kk) = t[tm];
    }
    // transposed orientation: line = column of W, contraction index = row
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * g + j] = tile[32 * grp + 16 * g + 4 * kk + j][line];
    pl_split8(v, t);
    if (c0 + line < jb.K && r0 + 32 * grp < jb.pt_ld && jb.pt) {
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
        *reinterpret_cast<pl_u4*>(jb.pt + tm * jb.pt_term + (int64_t)(c0 + line) * jb.pt_ld + r0 + 32 * grp + 8 * kk) = t[tm];
    }
  }
}

size_t planes_elems(int rows, int K) { return (size_t)3 * rows * round_up((size_t)K, 32); }

// Split up to four row-major weight matrices into planes (natural and / or transposed orientation; null = skip).
int planes_split(const PlaneSpec* specs, int n, hipStream_t st) {
  EP_REQUIRE(n >= 1 && n <= 4, EP_E_ARG, "planes_split: 1..4 matrices");
  PlaneJobs jobs{};
  jobs.n = n;
  int gx = 0, gy = 0;
  for (int i = 0; i < n; ++i) {
    const PlaneSpec& s = specs[i];
    EP_REQUIRE(s.W && s.R > 0 && s.K > 0 && (s.pn || s.pt), EP_E_ARG, "planes_split: bad matrix %d", i);
    EP_REQUIRE((!s.pn || aligned16(s.pn)) && (!s.pt || aligned16(s.pt)), EP_E_ALIGN, "planes_split: planes must be 16-byte aligned");
    PlaneJob& j = jobs.j[i];
    j.W = s.W; j.R = s.R; j.K = s.K; j.ldw = s.ldw;
    j.pn = s.pn; j.pn_ld = (int64_t)round_up((size_t)s.K, 32); j.pn_term = (int64_t)s.R * j.pn_ld;
    j.pt = s.pt; j.pt_ld = (int64_t)round_up((size_t)s.R, 32); j.pt_term = (int64_t)s.K * j.pt_ld;
    gx = gx > (s.K + 63) / 64 ? gx : (s.K + 63) / 64;
    gy = gy > (s.R + 63) / 64 ? gy : (s.R + 63) / 64;
  }
  hipLaunchKernelGGL(ep_planes_split_kernel, dim3(gx, gy, n), dim3(256), 0, st, jobs);
  EP_LAUNCH_CHECK("ep_planes_split_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// C[z][m][n] (+)= alpha * sum_k A[z](m,k) * W[z](n,k) (+ bias[n]):  A fp32 (K contiguous), W as planes (K contiguous)
// 64 x 64 tile per workgroup, 8 waves as 4 (rows) x 2 (columns): a wave owns ONE 16-row block of A and TWO 16-column
// blocks of W.  K-tile 32 = one MFMA K.  LDS stage = fp32 A image (8 KiB, XOR-swizzled on the DMA source address like
// ep_gemm_dma_kernel) + 12 plane pieces of 1 KiB in MFMA lane order ([block][term][lane], 16 bytes per lane): 20 KiB.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PLG_STB = 8192 + 12 * 1024;            // bytes per ring stage

template <int NST>
__global__ __launch_bounds__(768) void ep_gemm_planes_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7 multiply, 8..11 only move data
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int z = blockIdx.z;
  const int nk = (p.K + BK - 1) / BK;
  // Barrier k (k = 0 .. nk): K-tile k has landed completely AND every multiply wave holds tile k-1 in registers, so the
  // stage of tile k-1 may be refilled.  Both roles pass the same nk + 1 barriers.

  if (w >= 8) {
    // ---- loader waves: 20 pieces of 1 KiB per K-tile (8 of the fp32 A image, 12 of the weight planes), 5 per wave.
    // An LDS-DMA instruction costs its issuing wave ~100 cycles: on the multiply waves that would sit in the per-tile
    // critical path (barrier -> issue -> reads -> split -> MFMA chain), here it runs beside it.
    const int lw = w - 8;
    const float* A = p.A + (int64_t)z * p.sAz;
    const uint16_t* Wp = p.Bpl + (int64_t)z * p.sBpz;
    // A: pieces lw, lw + 4 of the swizzled fp32 image (chunk c of row r in slot c ^ ((r >> 1) & 7))
    int64_t srcA[2]; int kqA[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int pos = (lw + 4 * jj) * 64 + lane;
      const int r = pos >> 3, q = pos & 7;
      kqA[jj] = q ^ ((r >> 1) & 7);
      int row = m0 + r; row = row < p.M ? row : p.M - 1;
      srcA[jj] = (int64_t)row * p.lda + 4 * kqA[jj];
    }
    // W: piece (blk, term) = lw + 4 jj: lane (i16, kk) copies the 16 bytes at row n0 + 16 blk + i16, position k0 + 8 kk
    int64_t srcW[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int pb = lw + 4 * jj, blk = pb / 3, term = pb - 3 * blk;
      int row = n0 + 16 * blk + i16; row = row < p.N ? row : p.N - 1;
      srcW[jj] = term * p.pl_term + (int64_t)row * p.ldbp + 8 * kk;
    }
    const bool ktail = (p.K % BK) != 0;
    auto issue = [&](int t) {                        // K-tile t (clamped to the last one) into stage t % NST
      const int tt = t < nk ? t : nk - 1;
      char* st = lds + (t % NST) * PLG_STB;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        int64_t oa = srcA[jj] + (int64_t)tt * BK;
        if (ktail && tt == nk - 1 && tt * BK + 4 * kqA[jj] >= p.K) oa = srcA[jj] - 4 * kqA[jj];   // past K: chunk 0 (zeroed later)
        __builtin_amdgcn_global_load_lds((pl_gptr_t)(A + oa), (pl_lds_ptr_t)(st + (lw + 4 * jj) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
        __builtin_amdgcn_global_load_lds((pl_gptr_t)(Wp + srcW[jj] + (int64_t)tt * BK),
                                         (pl_lds_ptr_t)(st + 8192 + (lw + 4 * jj) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t);
    for (int k = 0; k <= nk; ++k) {
      pl_dma_wait<5 * (NST - 2)>();                  // my pieces of tile k
      pl_barrier();
      issue(k + NST - 1);                            // into the stage of tile k-1
    }
    pl_dma_wait<0>();
    return;
  }

  // ---- multiply waves: 4 (rows) x 2 (columns); rows 16 wm .. +15, columns 32 wn .. +31 ----
  const int wm = w >> 1, wn = w & 1;
  float* C = p.C + (int64_t)z * p.sCz;
  const bool ktail = (p.K % BK) != 0;
  int fragA[2];
  {
    const int r = wm * 16 + i16;
    fragA[0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
    fragA[1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
  }
  const int fragW = 8192 + (2 * wn * 3) * 1024 + lane * 16;
  f4v acc[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
  f4v xs[2][2];                                      // [set][g]: fp32 fragment of A block wm
  pl_u4 bs[2][2][3];                                 // [set][block][term]: planes of W blocks 2 wn, 2 wn + 1
  auto read_tile = [&](int t, f4v (&x)[2], pl_u4 (&b3)[2][3]) {
    const char* st = lds + (t % NST) * PLG_STB;
    x[0] = *reinterpret_cast<const f4v*>(st + fragA[0]);
    x[1] = *reinterpret_cast<const f4v*>(st + fragA[1]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) b3[bi][tm] = *reinterpret_cast<const pl_u4*>(st + fragW + (bi * 3 + tm) * 1024);
  };
  auto compute = [&](int t, const f4v (&x)[2], const pl_u4 (&b3)[2][3]) {
    float v[8] = {x[0][0], x[0][1], x[0][2], x[0][3], x[1][0], x[1][1], x[1][2], x[1][3]};
    if (ktail && t == nk - 1) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = (t * BK + 16 * g + 4 * kk + j >= p.K) ? 0.f : v[4 * g + j];
    }
    pl_u4 a3[3];
    pl_split8(v, a3);
    // smallest terms first: lo x hi, hi x lo, mid x mid, then the 2^-8 pair, then hi x hi
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = pl_mfma(a3[2], b3[bi][0], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = pl_mfma(a3[0], b3[bi][2], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = pl_mfma(a3[1], b3[bi][1], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = pl_mfma(a3[1], b3[bi][0], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = pl_mfma(a3[0], b3[bi][1], acc[bi]);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) acc[bi] = pl_mfma(a3[0], b3[bi][0], acc[bi]);
  };
  pl_barrier();                                      // barrier 0: tile 0 landed
  read_tile(0, xs[0], bs[0]);
  // step it: barrier it+1 (tile it+1 landed; the reads of tile it have completed: lgkmcnt(0) in front of the barrier) ->
  // start the reads of tile it+1 into the other register set -> split + multiply tile it while they fly
#define EP_PL_STEP(IT, F)                                    \
  {                                                          \
    pl_barrier();                                            \
    read_tile((IT) + 1, xs[(F) ^ 1], bs[(F) ^ 1]);           \
    compute((IT), xs[F], bs[F]);                             \
  }
  int it = 0;
  for (; it + 1 < nk; it += 2) {
    EP_PL_STEP(it, 0)
    EP_PL_STEP(it + 1, 1)
  }
  if (it < nk) EP_PL_STEP(it, 0)
#undef EP_PL_STEP

  {
    int rb[2], cb[2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) { rb[bi] = m0 + wm * 16; cb[bi] = n0 + wn * 32 + bi * 16; }
    store_acc_blocks<2>(p, C, z, rb, cb, acc, kk, i16);
  }
}

bool gemm_planes_ok(const GemmParams& p) {
  return p.Bpl && aligned16(p.A) && aligned16(p.Bpl) && p.lda % 4 == 0 && p.sAz % 4 == 0 && p.ldbp % 32 == 0 &&
         p.pl_term % 8 == 0 && p.sBpz % 8 == 0 && p.M > 0 && p.N > 0 && p.K > 0;
}

template <int NST>
static void planes_launch(const GemmParams& p, int batch, hipStream_t st) {
  constexpr int lds = NST * PLG_STB;
  auto k = ep_gemm_planes_kernel<NST>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, batch);
  hipLaunchKernelGGL(k, grid, dim3(768), lds, st, p);
}

int gemm_planes(const GemmParams& p, int batch, hipStream_t st) {
  EP_REQUIRE(gemm_planes_ok(p), EP_E_ALIGN, "gemm_planes: operands must be 16-byte aligned (A: lda %% 4; planes: row stride %% 32)");
  static int nst = -1;
  if (nst < 0) { const char* e = getenv("EP_PLANES_NST"); nst = e ? atoi(e) : 5; }
  if (nst == 7) planes_launch<7>(p, batch, st);      // 140 KiB
  else if (nst == 4) planes_launch<4>(p, batch, st); // 80 KiB: two workgroups per CU
  else planes_launch<5>(p, batch, st);               // 100 KiB: one 12-wave workgroup per CU, 3 K-tiles in flight
  EP_LAUNCH_CHECK("ep_gemm_planes_kernel");
  return 0;
}

}  // namespace ep
