// Device-side building blocks of the vector-ALU streaming token passes (ep_pool_stream.hip, ep_pool_bwd2.hip): the
// LDS-DMA ring helpers, the packed-FMA partial scores, the cross-lane butterfly and the SGPR-broadcast pooling update.
#pragma once
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"

namespace ep {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int TB = 4;                 // tokens per butterfly mini-batch (one per 16-lane row)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LAZY_MAX_THR = 12.0f; // rescale only when a score exceeds the running max by this

// s_waitcnt vmcnt(n) with a runtime (wave-uniform) n -- used only at the head/tail of a stream
__device__ __forceinline__ void wait_vmcnt(int n) {
#define EP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    EP_W(0) EP_W(1) EP_W(2) EP_W(3) EP_W(4) EP_W(5) EP_W(6) EP_W(7) EP_W(8) EP_W(9)
    EP_W(10) EP_W(11) EP_W(12) EP_W(13) EP_W(14) EP_W(15) EP_W(16) EP_W(17) EP_W(18) EP_W(19)
    EP_W(20) EP_W(21) EP_W(22) EP_W(23) EP_W(24) EP_W(25) EP_W(26) EP_W(27) EP_W(28) EP_W(29)
    EP_W(30) EP_W(31) EP_W(32) EP_W(33) EP_W(34) EP_W(35) EP_W(36) EP_W(37) EP_W(38) EP_W(39)
    EP_W(40) EP_W(41) EP_W(42) EP_W(43) EP_W(44) EP_W(45) EP_W(46) EP_W(47) EP_W(48)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef EP_W
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate out of range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void ring_barrier() {
  // this wave's LDS reads of the previous tile have retired (their results were consumed) and
  // its own DMA pieces of the next tile have landed (counted vmcnt just before).
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Issue the LDS-DMA copy of one ring item: valid bytes [0, limit+16) of `src`; lanes past the end
// re-copy the last 16 bytes (finite duplicates that are never consumed as valid rows).
// AUX: cache-policy bits of the copy (EP_DMA_AUX = nt for once-read tokens; 16 = sc1 for rows another workgroup wrote
// inside this launch, ep_inpass.h)
template <int NW, int KDMA, int AUX = EP_DMA_AUX>
__device__ __forceinline__ void dma_rows(const char* src, unsigned limit, char* slot, int npiece, int w,
                                         unsigned lane16) {
#pragma unroll
  for (int j = 0; j < KDMA; ++j) {
    int pc = w + NW * j;                       // wave-uniform piece index
    pc = pc < npiece ? pc : npiece - 1;        // surplus instructions re-copy the last piece
    unsigned off = (unsigned)pc * 1024u + lane16;
    off = off < limit ? off : limit;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lds_ptr_t)(slot + pc * 1024), 16, 0, AUX);
  }
}

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// lane-local partial dot products of TB token rows with QW query rows (packed FMAs)
template <int QW, int KP>
__device__ __forceinline__ void partial_scores(const f4 (&w)[QW][KP], const f4 (&xv)[TB][KP], float (&part)[QW][TB]) {
  // k-outer: the QW*TB dot products advance together, so consecutive v_pk_fma_f32 are independent (a dependent pair
  // needs a wait state between them)
  f2 s[QW][TB];
#pragma unroll
  for (int j = 0; j < QW; ++j)
#pragma unroll
    for (int t = 0; t < TB; ++t) s[j][t] = w[j][0].xy * xv[t][0].xy;
#pragma unroll
  for (int j = 0; j < QW; ++j)
#pragma unroll
    for (int t = 0; t < TB; ++t) s[j][t] = fma2(w[j][0].zw, xv[t][0].zw, s[j][t]);
#pragma unroll
  for (int k = 1; k < KP; ++k) {
#pragma unroll
    for (int j = 0; j < QW; ++j)
#pragma unroll
      for (int t = 0; t < TB; ++t) s[j][t] = fma2(w[j][k].xy, xv[t][k].xy, s[j][t]);
#pragma unroll
    for (int j = 0; j < QW; ++j)
#pragma unroll
      for (int t = 0; t < TB; ++t) s[j][t] = fma2(w[j][k].zw, xv[t][k].zw, s[j][t]);
  }
#pragma unroll
  for (int j = 0; j < QW; ++j)
#pragma unroll
    for (int t = 0; t < TB; ++t) part[j][t] = s[j][t].x + s[j][t].y;
}

// butterfly reduction: on return u[q] holds, in every lane of row t (lanes 16t..16t+15), the
// 64-lane sum of part[q][t].
// sum over the 16 lanes of a row for TWO values at once, as v_add_f32 with the DPP modifier on its first source: one
// instruction per value and level.  (From `v += dpp(v)` on two values hipcc makes two v_mov_b32_dpp + one v_pk_add_f32
// per level: 12 vector instructions instead of 8.)  A DPP source written by the previous vector instruction needs two
// wait states: the other value's add and one s_nop provide them.
__device__ __forceinline__ void row16_sum2(float& a, float& b) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(a), "+v"(b));
}

template <int QW>
__device__ __forceinline__ void butterfly(const float (&part)[QW][TB], float (&u)[QW]) {
  float f[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    // fold32(a,b): lanes<32 <- a, lanes>=32 <- b.  fold16(r0,r1): rows <- [r0.lo, r1.lo, r0.hi, r1.hi]
    // rows [t0,t1,t2,t3]  <=  r0 = fold32(t0,t2), r1 = fold32(t1,t3)
    const float r0 = fold32(part[q][0], part[q][2]);
    const float r1 = fold32(part[q][1], part[q][3]);
    f[q] = fold16(r0, r1);
  }
  if constexpr (QW == 2) {
    row16_sum2(f[0], f[1]);
    u[0] = f[0]; u[1] = f[1];
  } else {
#pragma unroll
    for (int q = 0; q < QW; ++q) u[q] = row16_sum(f[q]);
  }
}

template <int QW, int KP, bool BF16>
__device__ __forceinline__ void load_rows(const char* tile, int rowbytes, const int (&coff)[KP], f4 (&xv)[TB][KP], int f16 = 0) {
#pragma unroll
  for (int t = 0; t < TB; ++t)
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (BF16) xv[t][k] = h16x4_to_f4(*reinterpret_cast<const uint2*>(tile + t * rowbytes + coff[k]), f16);
      else xv[t][k] = *reinterpret_cast<const f4*>(tile + t * rowbytes + coff[k]);
    }
}

template <int QW, int KP>
__device__ __forceinline__ void accumulate_rows(const float (&wrow)[QW], const f4 (&xv)[TB][KP], f4 (&acc)[QW][KP]) {
#pragma unroll
  for (int j = 0; j < QW; ++j)
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const float a = readlane_f(wrow[j], 16 * t);      // SGPR broadcast of weight (q_j, token t)
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[j][k] += a * xv[t][k];
    }
}


}  // namespace ep
