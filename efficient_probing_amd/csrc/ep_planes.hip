// DEFAULT since round 4 for the classifier's two contractions at every D (EP_GEMM_PLANES unset = mode 2; mode 1 -- all four
// critical-path contractions -- for D >= 2048; 0 = the exact-f32 kernels of ep_gemm.hip): the planes are written by the
// optimizer's update kernel (ep_optim.hip: tile_update_emit, ep_planes_dev.h), so a step needs no split launch.
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
// Measured on MI355X (rocprofv3 device durations of the 1024 x 1000 x 768 logits / 1024 x 768 x 1000 dz contractions; f32
// kernel 20.4 / 20.8 us):
//   round 2
//   1. both operands split in registers by every wave that multiplies them: 19.5 us -- bound by the split's vector
//      instructions (132 per wave and K-tile);
//   2. both operands split once per workgroup, the terms handed over through LDS plane images: 20.0 us (21.5 us with
//      32-row tiles, two workgroups per CU) -- the round trip adds 6 bytes per element of LDS traffic;
//   3. the weights arrive split (planes in global memory, written once per step), the activations are split in registers:
//      21.1 us with the multiply waves also issuing the DMA, 17.1 / 20.3 us with four loader waves and double-buffered
//      fragments.
//   round 3 (in-kernel cycle stamps per K-tile, and the same kernel with one kind of work left out at a time)
//   4. Two independent limits sat at the same ~890 cycles per K-tile, which is why no change to either alone moved the time:
//      a. the loaders: 888 cycles with the multiply waves idle, the same when every workgroup streams the SAME operands and
//         with an XCD-aware tile order -- not bandwidth.  The plane pieces were copied lane (i16, kk) <- row i16: a
//         quarter-wave touched 16 cache lines, 64 tag look-ups per instruction.  With four consecutive lanes on one row's
//         64 bytes (and the matching swizzle on the read side): 550 cycles.  Staging through registers
//         (global_load_dwordx4 + ds_write_b128, two LDS stages) instead of LDS-DMA: 705.
//      b. one multiply wave per SIMD: matrix instructions 370, split 124 (44 hand-placed instructions) and LDS reads 232
//         cycles ADD UP in one wave's instruction stream whatever their order -- split pinned two instructions behind each
//         MFMA (matrix instructions as volatile asm: 866 cycles between barriers; as builtins held by asm pins, which cost
//         an s_nop each: 907), compiler order (884).
//   5. THIS file: two multiply waves per SIMD on alternate K-tiles, so that one wave's matrix instructions run beside the
//      other wave's reads and split: 16.4 / 18.8 us with the split in C (60 instructions, sunk or pinned), 14.7 / 16.7 us
//      with the split as 44 volatile single instructions.  Per own K-tile a multiply wave now spends 640 cycles reading and
//      splitting and 485 in its 24 matrix instructions (384 if nothing else issued): 730 cycles per K-tile.  Moving the lo
//      term's 12 instructions into the matrix phase: 15.1 / 17.1 us (worse).
// In the whole train step at 1024 x 256 x 768 the planes still do not pay: mode 2 (logits and dz only, the planes of Wc
// split on the aux stream beside the first token pass) 0.442 against 0.437 ms -- the two kernels save 10 us, the split beside
// the HBM-bound pass costs it 12-20 us; mode 1 0.460 ms.  Kept as an opt-in with its tests.
//
// The contraction kernel
//   * streams the fp32 activation tile AND the weight planes into LDS by LDS-DMA (20 KiB per K-tile), issued by four
//     loader waves;
//   * splits only its own activation fragment in registers;
//   * reads the weight operands ready-made (ds_read_b128, no vector work, no LDS write-back);
//   * issues 6 v_mfma_f32_16x16x32_bf16 per 16x16x32 block pair.
//
// Plane format (ep_planes_split_kernel): for a row-major matrix W (R x K) the three terms hi / mid / lo of every element
// as bf16, [term][row][Kp] with Kp = K rounded up to 32 (zero padded), the k-order PERMUTED inside every group of 32:
// position 8 kk + 4 g + j holds k = 16 g + 4 kk + j.  That is the k-assignment the activation fragment reads deliver
// (lane group kk of the MFMA reads chunk 4 g + kk of the swizzled fp32 image), so lane (i, kk) of a block finds its
// eight weight values as 16 contiguous bytes -- one DMA lane, one ds_read_b128 lane.  The transposed orientation
// (planes of W^T, for the two backward contractions that sum over W's row index) is produced in the same launch through
// an LDS transpose.
#include "ep_planes_dev.h"
#include "ep_sidetask.h"

namespace ep {

// ---------------------------------------------------------------------------------------------------------------------
// weights -> planes, both orientations.  One 64 x 64 tile of W per workgroup (256 threads), blockIdx.z = matrix.
// (PlaneJob and the tile -> planes body live in ep_planes_dev.h: the optimizer's update kernel emits planes with them.)
// ---------------------------------------------------------------------------------------------------------------------
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
  pl_emit_tile(jb, tile, r0, c0, tid);
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
// dst (C x R, leading dimension ldd) = src (R x C, leading dimension lds_)^T -- the fp32 operand of a weight-gradient
// contraction (sum over the batch index) in the K-contiguous form the planes kernel reads.  64 x 64 tiles through LDS.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ep_transpose_kernel(const float* __restrict__ src, int R, int C, int64_t lds_,
                                                           float* __restrict__ dst, int64_t ldd) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + 256 * i, r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? src[(int64_t)(r0 + r) * lds_ + c0 + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + 256 * i, c = e >> 6, r = e & 63;
    if (c0 + c < C && r0 + r < R) dst[(int64_t)(c0 + c) * ldd + r0 + r] = tile[r][c];
  }
}
int transpose_f32(const float* src, int R, int C, int64_t lds_, float* dst, int64_t ldd, hipStream_t st) {
  EP_REQUIRE(src && dst && R > 0 && C > 0 && lds_ >= C && ldd >= R, EP_E_ARG, "transpose_f32: bad argument");
  hipLaunchKernelGGL(ep_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, st, src, R, C, lds_, dst, ldd);
  EP_LAUNCH_CHECK("ep_transpose_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// C[z][m][n] (+)= alpha * sum_k A[z](m,k) * W[z](n,k) (+ bias[n]):  A fp32 (K contiguous), W as planes (K contiguous)
// 64 x 64 tile per workgroup, K-tile 32 = one MFMA K.  12 waves: 4 loaders + 8 multiply waves, two per SIMD, as
// 4 (16-row blocks of A) x 2 (K-tile parity): a multiply wave owns one 16-row block of A against all four 16-column blocks of
// W on every other K-tile; the two partial sums of a block meet through LDS at the end.
// LDS stage = fp32 A image (8 KiB, XOR-swizzled on the DMA source address like ep_gemm_dma_kernel) + 12 plane pieces of 1 KiB
// ([block][term], 16 rows x 64 bytes each): 20 KiB; five stages.
// ---------------------------------------------------------------------------------------------------------------------
// NB = 16-column blocks of W per workgroup: 4 (64 x 64 tile, five stages of 20 KiB) or 8 (64 x 128 tile, four stages of
// 32 KiB) -- the wide tile halves the activation bytes and the split per matrix instruction: 48 instead of 24 matrix
// instructions stand against one fragment split and 26 instead of 14 LDS reads; for contractions with enough 64 x 128 tiles
// to fill the chip twice (planes_wide).
// NT = weight terms multiplied: 3 (fp32 accuracy: six products per block) or 1 (the AMP-bf16 arithmetic mode, GemmParams.nterms:
// bf16(a) x bf16(w) with fp32 accumulation -- the hi plane IS the weight rounded to bf16, the activation fragment is rounded by
// the first four instructions of the split; one product per block, a third of the plane bytes).
template <int NB, int NT = 3> struct PlGeom {
  static constexpr int npw = NB * NT;                // 1-KiB weight pieces per K-tile ([block][term])
  static constexpr int stb = 8192 + npw * 1024;      // bytes per ring stage
  static constexpr int nst = (NT == 3 && NB == 8) ? 4 : 5;   // ring stages (128 / 100 KiB; single term: 80 / 60 KiB)
  static constexpr int npc = (8 + npw) / 4;          // 1-KiB pieces per loader wave and K-tile (8 / 5; single term 4 / 3)
};

// The split of one 8-value fragment as 44 single instructions (11 per value pair, pair index fastest so that neighbours are
// independent), each a volatile asm statement.  Per pair (x0, x1):
//   0 hi = cvt_pk(x0, x1) | 1, 2 the fp32 images of hi | 3, 4 r = x - image (exact) | 5 mid = cvt_pk(r) | 6, 7 images |
//   8, 9 s = r - image (exact, <= 8 bits left) | 10 lo = cvt_pk(s)
// Why asm: from the C form (pl_split8) hipcc makes 60 instructions with packed subtractions (v_pk_add_f32 beside matrix
// instructions costs ~13 cycles more than two v_sub_f32, MI355X_MICROARCH.md) and, the values being pure, sinks them across the
// workgroup barrier to the matrix instructions that use them -- the half of the K-tile step they are meant to stay out of.
struct PlSplit { float v[8]; float t[8]; };
template <int U>
__device__ __forceinline__ void pl_split_uop(PlSplit& s, const f4v (&x)[2], pl_u4 (&a)[3]) {
  constexpr int q = U & 3, o = U >> 2, e0 = 2 * q, e1 = 2 * q + 1, g = q >> 1, j0 = e0 & 3;
  if constexpr (o == 0 || o == 5 || o == 10) {
    unsigned pk;
    if constexpr (o == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[g][j0]), "v"(x[g][j0 + 1]));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(s.v[e0]), "v"(s.v[e1]));
    a[o / 5][q] = pk;
  } else if constexpr (o == 1 || o == 6) {
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.t[e0]) : "v"(a[o / 5][q]));
  } else if constexpr (o == 2 || o == 7) {
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.t[e1]) : "v"(a[o / 5][q]));
  } else if constexpr (o == 3) {
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s.v[e0]) : "v"(x[g][j0]), "v"(s.t[e0]));
  } else if constexpr (o == 4) {
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s.v[e1]) : "v"(x[g][j0 + 1]), "v"(s.t[e1]));
  } else if constexpr (o == 8) {
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[e0]) : "v"(s.t[e0]));
  } else {
    static_assert(o == 9, "11 instructions per pair");
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[e1]) : "v"(s.t[e1]));
  }
}
template <int U0, int U1>
__device__ __forceinline__ void pl_split_uops(PlSplit& s, const f4v (&x)[2], pl_u4 (&a)[3]) {
  if constexpr (U0 < U1) { pl_split_uop<U0>(s, x, a); pl_split_uops<U0 + 1, U1>(s, x, a); }
}

// The tile body: one 64 x (16 NB) output tile (mt_, nt_) of batch entry z_, all 12 waves of the workgroup (the loader waves
// and the odd-parity multiply waves leave through `return`: the callers do nothing behind it).  NST = ring stages.
template <int NB, int NT, int NST>
__device__ __forceinline__ void pl_tile_body(const GemmParams& p, char* lds, int mt_, int nt_, int z_) {
  constexpr int PLG_STB = PlGeom<NB, NT>::stb, NPC = PlGeom<NB, NT>::npc;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7 multiply, 8..11 only move data
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = mt_ * 64, n0 = nt_ * (16 * NB);
  const int z = z_;
  const int nk = (p.K + BK - 1) / BK;
  const bool ktail = (p.K % BK) != 0;
  // Barrier k (k = 0 .. nk): K-tile k has landed completely AND the reads of tile k-1 are done, so its stage may be refilled.
  // Every wave passes the same nk + 1 barriers, then F1 (the ring is quiet) and -- the multiply waves -- F2 (partial sums stored).

  if (w >= 8) {
    // ---- loader waves: 20 pieces of 1 KiB per K-tile (8 of the fp32 A image, 12 of the weight planes), 5 LDS-DMA per wave
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
    // W: piece (blk, term) = lw + 4 jj, 16 rows x 64 bytes.  Lane L fills chunk L of the piece: row L >> 2, and of that row's
    // four 16-byte k-chunks the one a multiply lane (i16, kk) looks for in slot kk ^ 2 (i16 >> 3) (conflict-free for the lane
    // groups of ds_read_b128).  Four CONSECUTIVE lanes read one row's 64 contiguous bytes: a quarter-wave touches 4 cache
    // lines.  (Round 2: lane (i16, kk) copied row i16 -- 16 lines per quarter-wave, 64 tag look-ups per instruction; that, not
    // bandwidth, made the loaders alone take 888 cycles per K-tile, the same with every workgroup streaming the SAME operands;
    // 550 now.)
    int64_t srcW[NPC - 2];
#pragma unroll
    for (int jj = 0; jj < NPC - 2; ++jj) {
      const int pb = lw + 4 * jj, blk = pb / NT, term = pb - NT * blk;
      const int r16 = lane >> 2, kq = (lane & 3) ^ (2 * (r16 >> 3));
      int row = n0 + 16 * blk + r16; row = row < p.N ? row : p.N - 1;
      srcW[jj] = term * p.pl_term + (int64_t)row * p.ldbp + 8 * kq;
    }
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
      for (int jj = 0; jj < NPC - 2; ++jj)
        __builtin_amdgcn_global_load_lds((pl_gptr_t)(Wp + srcW[jj] + (int64_t)tt * BK),
                                         (pl_lds_ptr_t)(st + 8192 + (lw + 4 * jj) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t);
    for (int k = 0; k <= nk; ++k) {
      pl_dma_wait<NPC * (NST - 2)>();                // my pieces of tile k
      pl_barrier();
      issue(k + NST - 1);                            // into the stage of tile k-1
    }
    pl_dma_wait<0>();
    pl_barrier();                                    // F1: the ring is quiet, its first NB x 4 KiB serve the final reduction
    return;
  }

  // ---- multiply waves: rows 16 wm .. +15, all four column blocks, K-tiles of parity `par` ----
  const int wm = w & 3, par = w >> 2;
  float* C = p.C + (int64_t)z * p.sCz;
  int fragA[2];
  {
    const int r = wm * 16 + i16;
    fragA[0] = r * 128 + 16 * ((0 + kk) ^ ((r >> 1) & 7));
    fragA[1] = r * 128 + 16 * ((4 + kk) ^ ((r >> 1) & 7));
  }
  const int fragW = 8192 + (4 * i16 + (kk ^ (2 * (i16 >> 3)))) * 16;
  f4v acc[NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) acc[bi] = f4v{0.f, 0.f, 0.f, 0.f};
  f4v xs[2];                                         // [g]: fp32 fragment of A block wm
  pl_u4 a3[3];                                       // [term]: its split
  pl_u4 bs[NB][NT];                                  // [block][term]: planes of the W blocks
  PlSplit sp;
  // Between barriers k and k+1 the waves of tile k's parity read it and split their fragment; between k+1 and k+2 they issue
  // its 24 matrix instructions -- while the waves of the other parity read and split tile k+1.  One wave feeds a SIMD's matrix
  // pipe while the other wave's LDS reads and vector instructions issue beside it.  Inside ONE wave the three kinds of work
  // add up whatever the instruction order (370 matrix + 124 split + 232 LDS-read cycles per K-tile, measured by leaving each
  // out in turn; 866 cycles between barriers with the split hand-placed two instructions behind each MFMA).
  // (Measured and dropped on the 64 x 128 tile: the second half of the plane reads behind the split (as first written: 894
  // cycles of read + split per own K-tile) or all reads in front of it (912) make no difference, and odd row blocks splitting
  // first while the even ones read first -- two waves contending for the LDS at a time -- is slower: 975.)
  auto phase_read = [&](int t) {
    const char* st = lds + (t % NST) * PLG_STB;
    xs[0] = *reinterpret_cast<const f4v*>(st + fragA[0]);
    xs[1] = *reinterpret_cast<const f4v*>(st + fragA[1]);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int tm = 0; tm < NT; ++tm) bs[bi][tm] = *reinterpret_cast<const pl_u4*>(st + fragW + (bi * NT + tm) * 1024);
    if (ktail && t == nk - 1) {                      // zero the positions at or past K (last tile of a ragged K only)
      const int klim = p.K - t * BK;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[g][j] = (16 * g + 4 * kk + j >= klim) ? 0.f : xs[g][j];
    }
    pl_split_uops<0, NT == 3 ? 44 : 4>(sp, xs, a3);  // volatile: the split stays on this side of the barrier (single term: the four roundings)
  };
  auto phase_mfma = [&]() {
    // smallest terms first: lo x hi, hi x lo, mid x mid, then the 2^-8 pair, then hi x hi
    if constexpr (NT == 1) {
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) acc[bi] = pl_mfma(a3[0], bs[bi][0], acc[bi]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 6 * NB; ++i) {
      const int pr = i / NB, bi = i % NB;
      const int ta = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1, tb = pr == 0 ? 0 : pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
      acc[bi] = pl_mfma(a3[ta], bs[bi][tb < NT ? tb : 0], acc[bi]);
    }
  };
  unsigned long long t_b0 = 0, t_rd = 0, t_b1 = 0, t_mm = 0;   // diagnostic only (EP_PLANES_STAMP)
  const bool stamped = p.ablate == 77;
  int k = 0;
  const unsigned long long t_c0 = stamped ? __builtin_readcyclecounter() : 0ull, t_r0 = stamped ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (par) { pl_barrier(); k = 1; }                  // barrier 0 belongs to the other parity's first tile
  for (; k < nk; k += 2) {
    const unsigned long long c0 = stamped ? __builtin_readcyclecounter() : 0ull;
    pl_barrier();                                    // barrier k: tile k landed
    const unsigned long long c1 = stamped ? __builtin_readcyclecounter() : 0ull;
    phase_read(k);
    const unsigned long long c2 = stamped ? __builtin_readcyclecounter() : 0ull;
    pl_barrier();                                    // barrier k + 1: my reads are done, the stage may be refilled
    const unsigned long long c3 = stamped ? __builtin_readcyclecounter() : 0ull;
    phase_mfma();
    if (stamped) { const unsigned long long c4 = __builtin_readcyclecounter(); t_b0 += c1 - c0; t_rd += c2 - c1; t_b1 += c3 - c2; t_mm += c4 - c3; }
  }
  if (stamped && lane == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(p.skws) + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + w) * 6;
    d[0] = t_b0; d[1] = t_rd; d[2] = t_b1; d[3] = t_mm;
    d[4] = __builtin_readcyclecounter() - t_c0; d[5] = __builtin_amdgcn_s_memrealtime() - t_r0;   // whole K loop: cycles, 100 MHz ticks
  }
  if ((nk & 1) == par) pl_barrier();                 // both parities pass barriers 0 .. nk
  pl_barrier();                                      // F1: the loaders are done with the ring
  float* red = reinterpret_cast<float*>(lds) + (wm * NB * 64 + lane) * 4;
  if (par) {
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) *reinterpret_cast<f4v*>(red + bi * 256) = acc[bi];
  }
  pl_barrier();                                      // F2 (the loader waves have left)
  if (par) return;
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) acc[bi] += *reinterpret_cast<const f4v*>(red + bi * 256);
  int rb[NB], cb[NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) { rb[bi] = m0 + wm * 16; cb[bi] = n0 + bi * 16; }
  store_acc_blocks<NB>(p, C, z, rb, cb, acc, kk, i16);
}

template <int NB, int NT>
__global__ __launch_bounds__(768) void ep_gemm_planes_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  int mt_ = blockIdx.y, nt_ = blockIdx.x, z_ = blockIdx.z;
  if (p.m_fast) {
    // 1-D launch, M-tiles fastest, XCD-aware: workgroup L runs on XCD L % 8, so XCD c walks the contiguous range of tile
    // numbers [c per, (c + 1) per) -- the M-tiles of one weight tile are neighbours in time on ONE XCD's L2
    const int mtn = (p.M + 63) / 64, ntn = (p.N + 16 * NB - 1) / (16 * NB);
    const unsigned per = gridDim.x / 8u, L = blockIdx.x;
    const unsigned V = (L % 8u) * per + L / 8u;
    if (V >= (unsigned)mtn * (unsigned)ntn * (unsigned)p.zn) return;
    const unsigned r = V / (unsigned)mtn;
    mt_ = (int)(V % (unsigned)mtn); nt_ = (int)(r % (unsigned)ntn); z_ = (int)(r / (unsigned)ntn);    // batched: z slowest
  }
  pl_tile_body<NB, NT, PlGeom<NB, NT>::nst>(p, lds, mt_, nt_, z_);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same contraction with SIDE WORK in its launch (round 6): the weight-gradient contractions of a head step (dWc =
// dlogits^T z, dWv_q = dy_q^T P_q), the bias column sum and the statistics fold feed nothing before the optimizer.  Until
// round 5 they rode in the launch of the second token pass (extra workgroups behind the pooling grid), where they cost the
// HBM-bound pass 30 - 35 us (256 x 768: 173 - 197 us in the step against 143 us alone) -- or ran on aux streams beside it,
// which costs more.  The critical-path contractions between the passes are one latency-bound 64 x 64 tile per CU with a
// quarter of the chip idle: their launches carry the side work instead.  1-D grid: blocks [0, main) are the tiles of the
// contraction (N-tiles fastest, then M-tiles, then the batch), blocks behind them run `side` with their first four waves
// (ep_sidetask.h: 256-thread bodies; the other eight waves leave at once).  Four ring stages (80 KiB) so that two workgroups
// -- a tile and a side block, or two side blocks -- share a CU.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PLS_NST = 4;
template <int NB, int NT>
__global__ __launch_bounds__(768) void ep_gemm_planes_side_kernel(GemmParams p, SideTasks side) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int mtn = (p.M + 63) / 64, ntn = (p.N + 16 * NB - 1) / (16 * NB);
  const int main_blocks = mtn * ntn * p.zn;
  const int L = blockIdx.x;
  if (L >= main_blocks) {
    if (threadIdx.x >= 256) return;
    run_side_task<true>(side, L - main_blocks, lds);
    return;
  }
  const int r = L / ntn;
  pl_tile_body<NB, NT, PLS_NST>(p, lds, r % mtn, L % ntn, r / mtn);
}

bool gemm_planes_ok(const GemmParams& p) {
  return p.Bpl && aligned16(p.A) && aligned16(p.Bpl) && p.lda % 4 == 0 && p.sAz % 4 == 0 && p.ldbp % 32 == 0 &&
         p.pl_term % 8 == 0 && p.sBpz % 8 == 0 && p.M > 0 && p.N > 0 && p.K > 0;
}

template <int NB, int NT>
static void planes_launch(const GemmParams& p, int batch, hipStream_t st) {
  constexpr int lds = PlGeom<NB, NT>::nst * PlGeom<NB, NT>::stb;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)ep_gemm_planes_kernel<NB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  dim3 grid((p.N + 16 * NB - 1) / (16 * NB), (p.M + 63) / 64, batch);
  GemmParams q = p;
  q.zn = batch;
  // EP_PLANES_MFAST=0 / 1 forces the launch order.  Default: the caller's m_fast for single contractions; batched launches keep
  // the 3-D grid (measured at 196 x 4096, y and dP as 8 batches: the XCD-aware 1-D order 2.354 - 2.359 ms per step against 2.342)
  static int mfast_env = -2;
  if (mfast_env == -2) { const char* e = getenv("EP_PLANES_MFAST"); mfast_env = e ? atoi(e) : -1; }
  if (mfast_env >= 0) q.m_fast = mfast_env; else if (batch != 1) q.m_fast = 0;
  if (q.m_fast) grid = dim3(8u * (unsigned)(((size_t)grid.x * grid.y * batch + 7) / 8), 1, 1);
  hipLaunchKernelGGL((ep_gemm_planes_kernel<NB, NT>), grid, dim3(768), lds, st, q);
}
// 64 x 128 tiles when they fill the chip at least twice (EP_PLANES_WIDE=0 / 1 forces)
static bool planes_wide(const GemmParams& p, int batch) {
  static int force = -2;
  if (force == -2) { const char* e = getenv("EP_PLANES_WIDE"); force = e ? atoi(e) : -1; }
  if (force >= 0) return force != 0;
  const long wide = (long)((p.N + 127) / 128) * ((p.M + 63) / 64) * batch, narrow = (long)((p.N + 63) / 64) * ((p.M + 63) / 64) * batch;
  // ... or when the 64 x 64 tiles need a SECOND round of the chip that the wide ones do not (one 12-wave workgroup per CU): dz at
  // 1024 x 1152 (SigLIP2 SO400M) is 288 narrow tiles on 256 CUs -- 32 - 44 us in the step against 20 for the logits (rocprofv3)
  const long cus = cu_count();
  return wide >= 2 * 256 || (narrow > cus && wide <= cus);
}

template <int NB, int NT>
static void planes_side_launch(const GemmParams& p, int batch, const SideTasks& sd, hipStream_t st) {
  constexpr size_t ring = (size_t)PLS_NST * PlGeom<NB, NT>::stb;
  constexpr size_t side_lds = SIDE_LDS_BYTES > w3d_lds_bytes<64>() ? SIDE_LDS_BYTES : w3d_lds_bytes<64>();
  constexpr int lds = (int)(ring > side_lds ? ring : side_lds);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)ep_gemm_planes_side_kernel<NB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  GemmParams q = p;
  q.zn = batch; q.m_fast = 0;
  const int main_blocks = ((p.N + 16 * NB - 1) / (16 * NB)) * ((p.M + 63) / 64) * batch;
  hipLaunchKernelGGL((ep_gemm_planes_side_kernel<NB, NT>), dim3(main_blocks + sd.total), dim3(768), lds, st, q, sd);
}
// C[z] (+)= alpha * A[z] W[z]^T (+ bias) as gemm_planes, with the side tasks `sd` as extra workgroups of the same launch
int gemm_planes_side(const GemmParams& p, int batch, const SideTasks& sd, hipStream_t st) {
  EP_REQUIRE(gemm_planes_ok(p), EP_E_ALIGN, "gemm_planes_side: operands must be 16-byte aligned (A: lda %% 4; planes: row stride %% 32)");
  if (sd.total <= 0) return gemm_planes(p, batch, st);
  // EP_SIDE_DMA=0: the side contractions on the register-prefetch tile (gemm_tile_b3g) instead of the LDS-ring one
  static int dma_on = -1;
  if (dma_on < 0) { const char* e = getenv("EP_SIDE_DMA"); dma_on = e ? atoi(e) : 1; }
  SideTasks sq = sd;
  if (dma_on && sq.b3 == 1) {
    bool ok = true;
    for (int i = 0; i < sq.n_gemm; ++i) ok &= sq.g[i].K >= 64 && sq.g[i].lda % 4 == 0 && sq.g[i].ldb % 4 == 0;
    if (ok) sq.b3 = 2;
  }
  if (p.nterms == 1 || gemm_arith() == 1) planes_side_launch<4, 1>(p, batch, sq, st);
  else planes_side_launch<4, 3>(p, batch, sq, st);
  EP_LAUNCH_CHECK("ep_gemm_planes_side_kernel");
  return 0;
}

int gemm_planes(const GemmParams& p, int batch, hipStream_t st) {
  EP_REQUIRE(gemm_planes_ok(p), EP_E_ALIGN, "gemm_planes: operands must be 16-byte aligned (A: lda %% 4; planes: row stride %% 32)");
  static int stamp = -1;                             // diagnostic only (EP_PLANES_STAMP=1): cycles per phase of a K-tile to stderr
  if (stamp < 0) { const char* e = getenv("EP_PLANES_STAMP"); stamp = e ? atoi(e) : 0; }
  const int tw = planes_wide(p, batch) ? 128 : 64, nwg = ((p.N + tw - 1) / tw) * ((p.M + 63) / 64);
  if (stamp && batch == 1 && nwg <= 4096) {
    static unsigned long long* dbg = nullptr;
    static unsigned long long host[4096 * 8 * 6];
    if (!dbg) (void)hipMalloc(&dbg, sizeof(host));
    GemmParams q = p;
    q.skws = reinterpret_cast<float*>(dbg); q.ablate = 77;
    (void)hipMemsetAsync(dbg, 0, (size_t)nwg * 8 * 6 * sizeof(unsigned long long), st);
    if (planes_wide(p, 1)) planes_launch<8, 3>(q, 1, st); else planes_launch<4, 3>(q, 1, st);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(host, dbg, (size_t)nwg * 8 * 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nwg * 8; ++i) for (int j = 0; j < 6; ++j) a[j] += (double)host[6 * i + j];
    const double n = (double)nwg * 8 * (((p.K + 31) / 32) / 2.0);
    static int printed = 0;
    if (printed++ % 10 == 5)
      fprintf(stderr, "[EP_PLANES_STAMP] %d x %d x %d, per own K-tile of a multiply wave: barrier k %.0f, read+split %.0f, barrier k+1 %.0f, matrix %.0f cycles; K loop of a workgroup %.0f cycles = %.1f us (%.2f GHz), %d workgroups\n",
              p.M, p.N, p.K, a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / (nwg * 8.0), a[5] / (nwg * 8.0) / 100.0, a[4] / a[5] / 10.0, nwg);
    return 0;
  }
  if (planes_big_wanted(p, batch)) planes_big_launch(p, batch, st);        // ep_planes_big.hip: 128 x 128 tiles for the large contractions
  else if (p.nterms == 1 || gemm_arith() == 1) {                           // AMP-bf16: one product per block
    if (planes_wide(p, batch)) planes_launch<8, 1>(p, batch, st); else planes_launch<4, 1>(p, batch, st);
  }
  else if (planes_wide(p, batch)) planes_launch<8, 3>(p, batch, st);
  else planes_launch<4, 3>(p, batch, st);
  EP_LAUNCH_CHECK("ep_gemm_planes_kernel");
  return 0;
}

}  // namespace ep
