// The planes contraction (ep_planes.hip: fp32 activations x pre-split bf16 weight planes at fp32 accuracy) for LARGE problems:
// 128 x 128 tiles, one wave per SIMD, software-pipelined inside the wave.  Round 5.
//
// Why a second kernel: ep_gemm_planes_kernel was shaped for the launch-bound contractions of the 256 x 768 step (64 x 64 /
// 64 x 128 tiles, a barrier pair per K-tile).  At the 196 x 4096 head (BASELINE configs[4]) its contractions are 34 GFLOP each
// and it ran them at 0.13 - 0.16 TFLOP/us fp32-equivalent (y = P Wv^T: 262 us; reference poolings/ep.py:40 and its autograd):
//   * every one of its four 16-row waves reads ALL of the tile's weight operands from LDS: 26 KiB per wave and K-tile,
//     104 KiB per K-tile of a workgroup = 830 LDS cycles against 768 matrix cycles;
//   * a 64 x 128 tile streams 32 KiB per K-tile for 48 matrix instructions per wave, and the batched launch (one z per query
//     slice) had no XCD-aware tile order.
// Here a wave owns 64 x 64 of a 128 x 128 tile: 8 + 12 b128 reads, four fragment splits and 48 v_mfma_f32_32x32x16_bf16 per
// K-tile of 32 -- 80 KiB of LDS reads per K-tile and workgroup against 1536 matrix cycles, 40 KiB of DMA per K-tile for twice
// the products.  One wave per SIMD, so the overlap of matrix instructions with the next K-tile's reads and split is made INSIDE
// the wave: the split fragments are double-buffered (tile t+1 is read and split while the matrix instructions of tile t issue).
//
// What sets the speed (measured, 1024 x 4096 x 4096 fp32-equivalent = 206 GFLOP of bf16 products, 256 workgroups;
// tools/planes_probe.py, diagnostic builds EP_PBG_ABLATE / EP_PBG_CLK; MI355X_MICROARCH.md "vector-instruction ISSUE cost"):
//   * matrix instructions alone (no DMA, no reads, no split): 127 - 132 us, 1701 cycles per K-tile (35 per 32x32x16 with the
//     barrier) at the 1.9 - 2.05 GHz the chip holds under this load: the floor of this tile shape, 1.6 PFLOP/s;
//   * everything else ADDS: the split 28 us, the A-fragment reads 28, the DMA 26 - 35, the plane reads 10 -- 214 us in all,
//     2270 cycles per K-tile at 1.55 - 1.75 GHz (the clock falls with the extra work: with 64 workgroups on a quarter of the chip
//     the same cycle count runs at 2.3 GHz).  ep_gemm_planes_kernel: 215 us on this single contraction.
//   * a matrix instruction holds the SIMD's vector issue for 8 of its cycles; every other instruction of the wave adds its
//     own.  With the 16x16x32 shape (first version of this file) the 2.3 vector instructions per matrix instruction of the
//     split did not fit the 8 free cycles of a 16-cycle gap: 246 us.  The 32x32x16 shape has the same 8 held cycles per 32:
//     three times the room per product -- the shape used here;
//   * left to the scheduler, or steered with sched_group_barrier, the matrix instructions end up in runs of 15 - 60 with the
//     reads and the split serialised between them; so the mix is laid out BY HAND: one matrix instruction per group with a
//     full scheduling fence (sched_barrier(0)) behind it, and the split cut into three stages of 3 - 4 instructions;
//   * instruction diet that did NOT pay: scalar-base DMA addressing, opaque read bases (immediates instead of 19 vector adds),
//     staggered split stages without hazard padding: 2315 -> 2270 cycles; a 7-instruction split on v_dot2c_f32_bf16: slower
//     (see PbgPair); a fourth ring stage: equal.
// In the 196 x 4096 step (profiles/r05): y = P Wv^T (8 batches of 1024 x 512 x 4096) 262 -> 206 us, the step 2.342 -> 2.32 ms;
// the classifier's logits (64 tiles of 128 x 128: a quarter of the chip) 61 -> 140 us and dz (K = 1000) 58 -> 64, so
// planes_big_wanted() takes only contractions with >= 192 tiles and K >= 2048; dP (K = 512) runs beside the weight gradients
// on the other queue and is equal in both kernels (506 / 498 us).
// Tile order: 1-D launch, workgroup L -> XCD L % 8; XCD c walks a contiguous range of (z, N-tile, M-tile) numbers, M fastest:
// the workgroups that share a weight tile and the ones that share an activation tile are neighbours in time on ONE L2.
//
// Ring: NST stages of 40 KiB = fp32 A image (128 rows x 128 B, chunk c of row r in slot c ^ ((r >> 1) & 7)) + 24 plane pieces
// of 1 KiB ([16-row block][term], the piece format of ep_planes.hip).  Every wave issues 10 LDS-DMA instructions per K-tile,
// addressed as (uniform tile base) + (32-bit lane offset): no vector arithmetic per instruction.
// k-assignment: instruction s (0 / 1) of a K-tile, lane half kh, multiplies the eight k of 16-byte chunk kk = 2 s + kh of the
// plane row (k = 4 kk .. + 3 and 16 + 4 kk .. + 3) -- the same bytes the 16x16x32 form reads, lanes mapped 32 rows x 2 halves.
#include "ep_planes_dev.h"
#include <type_traits>
#include <utility>

namespace ep {

// NT = weight terms multiplied: 3 (fp32 accuracy) or 1 (the AMP-bf16 arithmetic mode: bf16(a) x bf16(w), the hi plane only)
template <int NT> struct PbgGeom {
  static constexpr int stb = 16384 + 8 * NT * 1024;  // bytes per ring stage: fp32 A image + 8 NT plane pieces
  static constexpr int npc = 4 + 2 * NT;             // DMA instructions per wave and K-tile
  static constexpr int nm = NT == 3 ? 48 : 8;        // matrix instructions per wave and K-tile
};
#ifndef EP_PBG_NST
#define EP_PBG_NST 3                                 // ring stages of 40 KiB (3 or 4)
#endif
#ifndef EP_PBG_CLK
#define EP_PBG_CLK 0                                 // diagnostic builds only: shader cycles and 100 MHz ticks of every wave's K loop -> stderr
#endif
#ifndef EP_PBG_ABLATE
#define EP_PBG_ABLATE 0                              // diagnostic builds only: 1 no split, 2 no plane reads, 4 no DMA, 8 no matrix instructions, 16 no A reads
#endif

typedef float pbg_f16v __attribute__((ext_vector_type(16)));

template <int... I, class F>
__device__ __forceinline__ void pbg_for_seq(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

__device__ __forceinline__ pbg_f16v pbg_mfma(pl_u4 a, pl_u4 b, pbg_f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pl_bf8, a), __builtin_bit_cast(pl_bf8, b), c, 0, 0, 0);
}

// The split of one value pair (x = h + m + l exactly, round-to-nearest terms: pl_split2 of ep_planes_dev.h) in three stages of
// 3 + 4 + 4 plain vector instructions.  (Tried: the residuals x - fp32(h) as v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) selector --
// one instruction instead of unpack + subtract, 7 per pair instead of 11, bit-identical over 2^28 random pairs per exponent
// range, tools/dot2_split_check.hip -- but the dot instruction issues at 8+ cycles beside matrix instructions: 2499 cycles per
// K-tile against 2315.  Also: written as constants the selectors are encoded as inline operands of the wrong half.)
struct PbgPair { float r0, r1; unsigned h, m; float t0, t1; };
__device__ __forceinline__ void pbg_stage0(PbgPair& s, float v0, float v1) {
  s.h = pl_pack_rne(v0, v1);
  s.t0 = __uint_as_float(s.h << 16); s.t1 = __uint_as_float(s.h & 0xffff0000u);
  s.r0 = v0; s.r1 = v1;
}
__device__ __forceinline__ void pbg_stage1(PbgPair& s) {
  s.r0 -= s.t0; s.r1 -= s.t1;                        // exact
  s.m = pl_pack_rne(s.r0, s.r1);
  s.t0 = __uint_as_float(s.m << 16);
}
__device__ __forceinline__ unsigned pbg_stage2(PbgPair& s) {
  s.t1 = __uint_as_float(s.m & 0xffff0000u);
  return pl_pack_rne(s.r0 - s.t0, s.r1 - s.t1);      // exact differences, <= 8 bits
}
// Instruction groups of the split stages of value pair p (16 pairs per K-tile) among the 48 groups: stage 0 in group 6 + 2 p,
// stage 1 three groups later, stage 2 two more: consecutive stages of one pair never share a group, so no stage waits for the
// result of the instruction in front of it.
constexpr int pbg_stage_group(int p, int stg) { return 6 + 2 * p + (stg == 0 ? 0 : stg == 1 ? 3 : 5); }

// one LDS-DMA instruction, 16 bytes per lane: global address = uniform 64-bit base + 32-bit lane offset (no vector arithmetic;
// through the builtin the compiler keeps ten 64-bit lane pointers and one v_lshl_add_u64 per instruction), LDS address in M0
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"       // (M0 is named as clobbered on purpose: the compiler sets it before each DMA of its own)
__device__ __forceinline__ void pbg_dma(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int NST, int NT, int WGS = 1>
__global__ __launch_bounds__(256, WGS) void ep_gemm_planes_big_kernel(GemmParams p, int mtn, int ntn, unsigned ntiles, int nfast) {
  constexpr int PBG_STB = PbgGeom<NT>::stb, PBG_NPC = PbgGeom<NT>::npc, NM = PbgGeom<NT>::nm;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i32 = lane & 31, kh = lane >> 5, i16 = lane & 15;
  const int wm = w & 1, wn = w >> 1;
  const unsigned per = gridDim.x / 8u, L = blockIdx.x;
  const unsigned V = (L % 8u) * per + L / 8u;
  if (V >= ntiles) return;
  // (the divisions run on the vector ALU: hand the results back to scalar registers, or every address built from them is a
  // vector value that has to be read back lane 0 by lane 0 for each DMA instruction)
  // nfast (planes_big_launch_nt): the N-tiles of one M-tile are neighbours -- the order for activations larger than the weights
  const int mt = __builtin_amdgcn_readfirstlane((int)(nfast ? (V / (unsigned)ntn) % (unsigned)mtn : V % (unsigned)mtn)),
            nt = __builtin_amdgcn_readfirstlane((int)(nfast ? V % (unsigned)ntn : (V / (unsigned)mtn) % (unsigned)ntn)),
            z = __builtin_amdgcn_readfirstlane((int)(V / (unsigned)(mtn * ntn)));
  const int m0 = mt * 128, n0 = nt * 128;
  const int nk = (p.K + BK - 1) / BK;
  const bool ktail = (p.K % BK) != 0;

  // ---- DMA sources: uniform tile bases + 32-bit lane offsets (bytes).  A pieces w, w+4, w+8, w+12 (8 rows x 128 B each);
  // plane pieces w + 4 jj, jj = 0..5 (16 rows x 64 B each)
  const char* Ab = reinterpret_cast<const char*>(p.A + (int64_t)z * p.sAz + (int64_t)m0 * p.lda);
  const char* Wb = reinterpret_cast<const char*>(p.Bpl + (int64_t)z * p.sBpz + (int64_t)n0 * p.ldbp);
  unsigned offA[4]; int kqA;
  {
    const int r8 = lane >> 3, q = lane & 7;
    kqA = q ^ ((4 * (w & 1) + (lane >> 4)) & 7);     // ((r >> 1) & 7) of row r = 8 (w + 4 jj) + r8: the same for every jj
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      int row = 8 * (w + 4 * jj) + r8; row = m0 + row < p.M ? row : p.M - 1 - m0;
      offA[jj] = (unsigned)(((int64_t)row * p.lda + 4 * kqA) * 4);
    }
  }
  unsigned offW[2 * NT];
  int64_t termW[2 * NT];                             // (uniform) term offsets, bytes
#pragma unroll
  for (int jj = 0; jj < 2 * NT; ++jj) {
    const int pb = w + 4 * jj, blk = pb / NT, term = pb - NT * blk;
    const int r16 = lane >> 2, kq = (lane & 3) ^ (2 * (r16 >> 3));
    int row = 16 * blk + r16; row = n0 + row < p.N ? row : p.N - 1 - n0;
    offW[jj] = (unsigned)(((int64_t)row * p.ldbp + 8 * kq) * 2);
    termW[jj] = (int64_t)term * p.pl_term * 2;
  }
  auto issue = [&](int t) {                          // K-tile t (< nk) into stage t % NST
    char* st = lds + (t % NST) * PBG_STB;
    const bool past = ktail && t == nk - 1;          // ragged last K-tile: chunks at or past K read chunk 0 (zeroed after the read)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const unsigned o = (past && t * BK + 4 * kqA >= p.K) ? offA[jj] - 16u * (unsigned)kqA : offA[jj] + (unsigned)t * (BK * 4);
      __builtin_amdgcn_global_load_lds((pl_gptr_t)(Ab + o), (pl_lds_ptr_t)(st + (w + 4 * jj) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 2 * NT; ++jj)
      __builtin_amdgcn_global_load_lds((pl_gptr_t)(Wb + termW[jj] + (int64_t)t * (BK * 2) + offW[jj]),
                                       (pl_lds_ptr_t)(st + 16384 + (w + 4 * jj) * 1024), 16, 0, 0);
  };

  // ---- fragment addresses.  A: row 64 wm + 32 mb + i32, fp32 chunks (2 s + kh) and 4 + (2 s + kh); planes: 16-row block
  // 4 wn + 2 nb + (i32 >> 4), term, chunk 2 s + kh
  int fragA[2][2];                                   // [s][g]; + mb * 4096
  int fragW[2];                                      // [s]; + nb * NBO + term * 1024
  constexpr int NBO = 2 * NT * 1024;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int r = wm * 64 + i32, c = 2 * s + kh, sw = (i32 >> 1) & 7;
    fragA[s][0] = r * 128 + 16 * (c ^ sw);
    fragA[s][1] = r * 128 + 16 * ((4 + c) ^ sw);
    fragW[s] = 16384 + (wn * 4 + (i32 >> 4)) * (NT * 1024) + (4 * i16 + (c ^ (2 * (i16 >> 3)))) * 16;
  }

  pbg_f16v acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  // Operand registers.  The split fragments of A are double-buffered.  Of the weight planes only the hi terms are: the
  // products are ordered by the weight term they use -- (hi, lo) | (mid, mid), (hi, mid) | (lo, hi), (mid, hi), (hi, hi) -- so
  // the lo planes of tile t+1 are read into the SAME registers after matrix instruction 7 and the mid planes after 23.
  // (The order of the six products inside a K-tile does not matter for the accuracy: from the second K-tile on the accumulator
  // holds the large partial sum anyway.)
  pl_u4 a3[2][2][2][NT];                             // [buffer][M block][s][term]
  pl_u4 bhi[2][2][2], bmid[2][2], blo[2][2];         // [N block][s]

  auto wait_tile = [&](int ahead) {                  // my DMA parts of a tile have landed, `ahead` = tiles issued behind it
    if (ahead <= 0) pl_dma_wait<0>();
    else if (ahead == 1) pl_dma_wait<PBG_NPC>();
    else if (ahead == 2) pl_dma_wait<2 * PBG_NPC>();
    else pl_dma_wait<3 * PBG_NPC>();
  };
  static_assert(NST == 3 || NST == 4, "ring depth");
  auto read_planes = [&](const char* st, pl_u4 (&b)[2][2], int term) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 2; ++s) b[nb][s] = *reinterpret_cast<const pl_u4*>(st + fragW[s] + nb * NBO + term * 1024);
  };
  // fragments of A of the tile in stage `st`, zeroed at or past klim, split into a[mb][s][term] (plain form: prologue, tail)
  auto read_split_plain = [&](const char* st, int klim, pl_u4 (&a)[2][2][NT]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f4v x0 = *reinterpret_cast<const f4v*>(st + fragA[s][0] + mb * 4096), x1 = *reinterpret_cast<const f4v*>(st + fragA[s][1] + mb * 4096);
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (16 * (e >> 2) + 4 * (2 * s + kh) + (e & 3) >= klim) ? 0.f : v[e];
        if constexpr (NT == 3) pl_split8(v, a[mb][s]);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[mb][s][0][e] = pl_pack_rne(v[2 * e], v[2 * e + 1]);
        }
      }
  };
  // matrix instruction i of a K-tile on register buffer `cur`: phase i >> 3 = product, then s, M block, N block
  auto mfma_at = [&](auto cur_c, auto i_c) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, i = decltype(i_c)::value;
    constexpr int ph = NT == 3 ? i >> 3 : 5, s = (i >> 2) & 1, mb = (i >> 1) & 1, nb = i & 1;
    constexpr int ta = ph == 0 ? 0 : ph == 1 ? 1 : ph == 2 ? 0 : ph == 3 ? 2 : ph == 4 ? 1 : 0;
    if constexpr (NT == 1) acc[mb][nb] = pbg_mfma(a3[cur][mb][s][0], bhi[cur][nb][s], acc[mb][nb]);
    else if constexpr (EP_PBG_ABLATE & 8) { if constexpr (i < 4) acc[mb][nb][0] += __uint_as_float(a3[cur][mb][s][ta < NT ? ta : 0][0] ^ blo[nb][s][0] ^ bmid[nb][s][1] ^ bhi[cur][nb][s][2]); }
    else if constexpr (ph == 0) acc[mb][nb] = pbg_mfma(a3[cur][mb][s][ta < NT ? ta : 0], blo[nb][s], acc[mb][nb]);
    else if constexpr (ph <= 2) acc[mb][nb] = pbg_mfma(a3[cur][mb][s][ta < NT ? ta : 0], bmid[nb][s], acc[mb][nb]);
    else acc[mb][nb] = pbg_mfma(a3[cur][mb][s][ta < NT ? ta : 0], bhi[cur][nb][s], acc[mb][nb]);
  };

  // prologue: the whole ring in flight; tile 0 -> buffer 0
#pragma unroll
  for (int t = 0; t < NST; ++t)
    if (t < nk) issue(t);
  wait_tile(nk - 1 < NST - 1 ? nk - 1 : NST - 1);
  pl_barrier();
  read_split_plain(lds, (ktail && nk == 1) ? p.K : BK, a3[0]);
  read_planes(lds, bhi[0], 0);
  if constexpr (NT == 3) { read_planes(lds, bmid, 1); read_planes(lds, blo, 2); }

  // Iteration t: (barrier) tile t+1 has landed and the stage of tile t is free -> refill it with tile t + NST; read and split
  // tile t+1 while the matrix instructions of tile t issue.  STEADY: no branch in the body; group i = matrix instruction i
  //   + DMA instruction i (i < 10)
  //   + LDS reads: A fragments two per group (0..3), hi planes 4..7, lo planes 8..11 (their last use is instruction 7),
  //     mid planes 24..27 (last use 23)
  //   + the split stages pbg_stage_group() places here (16 value pairs x 3 stages).
  auto steady = [&](int t, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
    pl_dma_wait<(NST - 2) * PBG_NPC>();
    pl_barrier();
    const unsigned stio = (unsigned)(t % NST) * PBG_STB;      // refill: the stage tile t leaves (tile t + NST); LDS byte address
    const int sto = ((t + 1) % NST) * PBG_STB;
    const char* Ak = Ab + (int64_t)(t + NST) * (BK * 4);
    const char* Wk = Wb + (int64_t)(t + NST) * (BK * 2);
    // per-iteration read bases, made opaque so that the block / term offsets stay instruction immediates (left transparent,
    // the compiler hoists all 14 sums out of the loop and adds the stage offset to each of them in every iteration)
    int bA[2][2], bW[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      bA[s_][0] = sto + fragA[s_][0]; bA[s_][1] = sto + fragA[s_][1]; bW[s_] = sto + fragW[s_];
      asm volatile("" : "+v"(bA[s_][0]), "+v"(bA[s_][1]), "+v"(bW[s_]));
    }
    f4v xs[4][2];                                    // [fragment f = 2 mb + s][g]
    if constexpr (NT == 1) {
      // single product: 8 matrix instructions per K-tile.  Group i = matrix instruction i + DMA instruction i (i < 6) + the two
      // reads of A fragment i (i < 4) or one hi-plane read (i >= 4) + the four roundings of fragment i - 4 (i >= 4)
      pbg_for_seq(std::make_integer_sequence<int, 8>{}, [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        mfma_at(cur_c, ic);
        if constexpr (i < 4) pbg_dma(Ak, offA[i & 3], stio + (w + 4 * (i & 3)) * 1024);
        else if constexpr (i < 6) pbg_dma(Wk + termW[i - 4], offW[i - 4], stio + 16384 + (w + 4 * (i - 4)) * 1024);
        if constexpr (i < 4) {
          xs[i][0] = *reinterpret_cast<const f4v*>(lds + bA[i & 1][0] + (i >> 1) * 4096);
          xs[i][1] = *reinterpret_cast<const f4v*>(lds + bA[i & 1][1] + (i >> 1) * 4096);
        } else {
          bhi[nxt][(i - 4) >> 1][(i - 4) & 1] = *reinterpret_cast<const pl_u4*>(lds + bW[(i - 4) & 1] + ((i - 4) >> 1) * NBO);
          constexpr int f = i - 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) a3[nxt][f >> 1][f & 1][0][e] = pl_pack_rne(xs[f][e >> 1][2 * (e & 1)], xs[f][e >> 1][2 * (e & 1) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
    PbgPair pr[4];                                   // pairs in flight
    pbg_for_seq(std::make_integer_sequence<int, 48>{}, [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      mfma_at(cur_c, ic);
      if constexpr (EP_PBG_ABLATE & 4) {}
      else if constexpr (i < 4) pbg_dma(Ak, offA[i & 3], stio + (w + 4 * (i & 3)) * 1024);
      else if constexpr (i < 10) pbg_dma(Wk + termW[(i - 4) % (2 * NT)], offW[(i - 4) % (2 * NT)], stio + 16384 + (w + 4 * ((i - 4) % 6)) * 1024);
      if constexpr (i < 4) {                         // fragment f = i: (mb = i >> 1, s = i & 1)
        if constexpr (EP_PBG_ABLATE & 16) { xs[i][0] = f4v{1.f, 2.f, 3.f, 4.f}; xs[i][1] = f4v{5.f, 6.f, 7.f, 8.f}; }
        else {
          xs[i][0] = *reinterpret_cast<const f4v*>(lds + bA[i & 1][0] + (i >> 1) * 4096);
          xs[i][1] = *reinterpret_cast<const f4v*>(lds + bA[i & 1][1] + (i >> 1) * 4096);
        }
      } else if constexpr (EP_PBG_ABLATE & 2) {}
      else if constexpr (i < 8) bhi[nxt][(i - 4) >> 1][(i - 4) & 1] = *reinterpret_cast<const pl_u4*>(lds + bW[(i - 4) & 1] + ((i - 4) >> 1) * NBO + 0 * 1024);
      else if constexpr (i < 12) blo[(i - 8) >> 1][(i - 8) & 1] = *reinterpret_cast<const pl_u4*>(lds + bW[(i - 8) & 1] + ((i - 8) >> 1) * NBO + 2 * 1024);
      else if constexpr (i >= 24 && i < 28) bmid[(i - 24) >> 1][(i - 24) & 1] = *reinterpret_cast<const pl_u4*>(lds + bW[(i - 24) & 1] + ((i - 24) >> 1) * NBO + 1 * 1024);
      // split stages placed in this group
      pbg_for_seq(std::make_integer_sequence<int, 48>{}, [&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value, pair = q / 3, stg = q % 3;
        if constexpr (pbg_stage_group(pair, stg) == i) {
          constexpr int f = pair >> 2, e = pair & 3;                     // elements 2e, 2e+1 of fragment f
          PbgPair& ps = pr[pair & 3];
          if constexpr (EP_PBG_ABLATE & 1) {
            if constexpr (stg == 2) { a3[nxt][f >> 1][f & 1][0][e] = __float_as_uint(xs[f][e >> 1][2 * (e & 1)]); a3[nxt][f >> 1][f & 1][1][e] = __float_as_uint(xs[f][e >> 1][2 * (e & 1) + 1]); a3[nxt][f >> 1][f & 1][2][e] = 0x3f803f80u; }
          } else if constexpr (stg == 0) pbg_stage0(ps, xs[f][e >> 1][2 * (e & 1)], xs[f][e >> 1][2 * (e & 1) + 1]);
          else if constexpr (stg == 1) pbg_stage1(ps);
          else {
            const unsigned l = pbg_stage2(ps);
            a3[nxt][f >> 1][f & 1][0][e] = ps.h; a3[nxt][f >> 1][f & 1][1][e] = ps.m; a3[nxt][f >> 1][f & 1][2][e] = l;
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    }
  };
  // the last iterations: the same data flow in plain blocks (waits by the number of tiles still behind, no refill past the end,
  // the ragged last K-tile zeroed before its split)
  auto tail = [&](int t, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
    const bool more = t + 1 < nk;
    const char* st = lds + ((t + 1) % NST) * PBG_STB;
    if (more) {
      wait_tile((nk - 1 < t + NST - 1 ? nk - 1 : t + NST - 1) - (t + 1));
      pl_barrier();
      if (t + NST < nk) issue(t + NST);
      read_planes(st, bhi[nxt], 0);
    }
    pbg_for_seq(std::make_integer_sequence<int, 8>{}, [&](auto ic) __attribute__((always_inline)) { mfma_at(cur_c, std::integral_constant<int, decltype(ic)::value>{}); });
    if constexpr (NT == 3) {
      if (more) read_planes(st, blo, 2);
      pbg_for_seq(std::make_integer_sequence<int, 16>{}, [&](auto ic) __attribute__((always_inline)) { mfma_at(cur_c, std::integral_constant<int, 8 + decltype(ic)::value>{}); });
      if (more) read_planes(st, bmid, 1);
      pbg_for_seq(std::make_integer_sequence<int, 24>{}, [&](auto ic) __attribute__((always_inline)) { mfma_at(cur_c, std::integral_constant<int, 24 + decltype(ic)::value>{}); });
    }
    if (more) read_split_plain(st, (ktail && t + 1 == nk - 1) ? p.K - (t + 1) * BK : BK, a3[nxt]);
  };
  int t = 0;
#if EP_PBG_CLK
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int nsteady = nk - NST - (ktail ? 1 : 0);    // iterations whose refill tile is a full one
  for (; t + 1 < nsteady; t += 2) {                  // pairs of steady iterations
    steady(t, std::integral_constant<int, 0>{});
    steady(t + 1, std::integral_constant<int, 1>{});
  }
  for (; t < nk; t += 2) {
    tail(t, std::integral_constant<int, 0>{});
    if (t + 1 >= nk) break;
    tail(t + 1, std::integral_constant<int, 1>{});
  }
  pl_dma_wait<0>();
#if EP_PBG_CLK
  if (lane == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(p.skws) + ((size_t)V * 4 + w) * 2;
    d[0] = __builtin_readcyclecounter() - clk_c0; d[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif

  // ---- epilogue: acc[mb][nb][r] = C[64 wm + 32 mb + 8 (r >> 2) + 4 kh + (r & 3)][64 wn + 32 nb + i32]
  float* C = p.C + (int64_t)z * p.sCz;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = n0 + wn * 64 + nb * 32 + i32;
    const bool cok = col < p.N;
    const int colc = cok ? col : p.N - 1;
    const float bv = p.bias ? p.bias[(int64_t)z * p.sBiasz + colc] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float v[16];
      int64_t off[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mb * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
        off[r] = (int64_t)(row < p.M ? row : p.M - 1) * p.ldc + colc;
        v[r] = p.alpha * acc[mb][nb][r] + bv;
      }
      if (p.accumulate) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = C[off[r]];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += old[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (cok && m0 + wm * 64 + mb * 32 + 8 * (r >> 2) + 4 * kh + (r & 3) < p.M) C[off[r]] = v[r];
    }
  }
}

// big tiles when the contraction is long enough to amortise a 40 KiB-per-K-tile ring and has enough 128 x 128 tiles to give
// most CUs one (EP_PLANES_BIG=0 / 1 forces)
bool planes_big_wanted(const GemmParams& p, int batch) {
  static int force = -2;
  if (force == -2) { const char* e = getenv("EP_PLANES_BIG"); force = e ? atoi(e) : -1; }
  if (force >= 0) return force != 0;
  const long tiles = (long)((p.N + 127) / 128) * ((p.M + 127) / 128) * batch;
  if (p.nterms == 1 || gemm_arith() == 1) {          // single product: bandwidth-bound -- the larger tile halves the operand traffic
    static int amp_big = -1;
    if (amp_big < 0) { const char* e = getenv("EP_PLANES_BIG_AMP"); amp_big = e ? atoi(e) : 1; }
    return amp_big && tiles >= 192 && p.K >= 256;
  }
  // ... and the token-matrix contractions of the matrix-core-bound heads (65536 rows: >= 2048 tiles): 1167 -> 1099 us at
  // 65536 x 1152 x 1152, 543 -> 501 us at N = 512, 2034 -> 1942 us at 65536 x 768 x 3072; 288 tiles of K = 1152: 90 -> 104 us (not taken)
  return (tiles >= 192 && p.K >= 2048) || (tiles >= 2048 && p.K >= 512);
}

template <int NT, int WGS>
static void planes_big_launch_nt(const GemmParams& p, int batch, hipStream_t st) {
  constexpr int NST = NT == 1 ? (WGS == 2 ? 3 : 4) : EP_PBG_NST;   // (single term: 24 KiB stages; two workgroups per CU: three each)
  constexpr int lds = NST * PbgGeom<NT>::stb;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)ep_gemm_planes_big_kernel<NST, NT, WGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  const int mtn = (p.M + 127) / 128, ntn = (p.N + 127) / 128;
  const unsigned ntiles = (unsigned)mtn * (unsigned)ntn * (unsigned)batch;
  const unsigned grid = 8u * ((ntiles + 7u) / 8u);
  // Tile order inside an XCD's range.  M fastest (round 5) keeps one WEIGHT tile in the XCD's L2 while the activations stream
  // past it: right for the EP step (1024 rows against 4096 x 4096 weights).  With 65536 token rows against a 1152-wide layer
  // (AbMILP, the DINOv2 block) it makes every XCD read ALL activations once per N-tile -- 9 / 27 x 302 MB per contraction from
  // HBM; N fastest reads an activation tile once and its other N-tiles hit the L2 (the weights, <= 8 MB, stay resident).
  // Taken when the activations are at least four times the weights.  EP_PBG_ORDER=0 / 1 forces M / N fastest (A/B runs).
  static int order = -2;
  if (order == -2) { const char* e = getenv("EP_PBG_ORDER"); order = e ? atoi(e) : -1; }
  const int nfast = order >= 0 ? order : ((double)p.M * 4.0 >= 4.0 * (double)p.N * 2.0 * NT && ntn > 1) ? 1 : 0;
#if EP_PBG_CLK
  static unsigned long long* dbg = nullptr;
  static unsigned long long host[8192 * 8];
  if (!dbg) (void)hipMalloc(&dbg, sizeof(host));
  GemmParams q = p;
  q.skws = reinterpret_cast<float*>(dbg);
  if (ntiles <= 8192) {
    hipLaunchKernelGGL((ep_gemm_planes_big_kernel<NST, NT, WGS>), dim3(grid), dim3(256), lds, st, q, mtn, ntn, ntiles, nfast);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(host, dbg, (size_t)ntiles * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (unsigned i = 0; i < ntiles * 4; ++i) { c += (double)host[2 * i]; r += (double)host[2 * i + 1]; }
    static int printed = 0;
    if (printed++ % 8 == 7)
      fprintf(stderr, "[EP_PBG_CLK] %d x %d x %d: K loop of a wave %.0f cycles = %.1f us: %.2f GHz, %.0f cycles per K-tile\n", p.M, p.N, p.K,
              c / (ntiles * 4.0), r / (ntiles * 4.0) / 100.0, c / r / 10.0, c / (ntiles * 4.0) / ((p.K + 31) / 32));
    return;
  }
#endif
  hipLaunchKernelGGL((ep_gemm_planes_big_kernel<NST, NT, WGS>), dim3(grid), dim3(256), lds, st, p, mtn, ntn, ntiles, nfast);
}
void planes_big_launch(const GemmParams& p, int batch, hipStream_t st) {
  // single-term form: its K-tile is 8 matrix instructions per wave against the same DMA issue, fragment reads and barrier as the
  // three-term form's 48 -- one wave per SIMD leaves the matrix pipe idle most of the time, so TWO workgroups share a CU
  // (212 VGPRs, 72 KiB of LDS each).  EP_PBG_AMP_WGS=1: one workgroup with a four-stage ring (round 5).
  static int wgs = -1;
  if (wgs < 0) { const char* e = getenv("EP_PBG_AMP_WGS"); wgs = e ? atoi(e) : 2; }
  if (p.nterms == 1 || gemm_arith() == 1) { if (wgs == 2) planes_big_launch_nt<1, 2>(p, batch, st); else planes_big_launch_nt<1, 1>(p, batch, st); }
  else planes_big_launch_nt<3, 1>(p, batch, st);
}

}  // namespace ep
