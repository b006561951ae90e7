// Device-side pieces of the bf16-plane format shared by ep_planes.hip (the split launch, the contraction kernel) and
// ep_optim.hip (the optimizer's update kernel writes the planes of the weight matrices it has just updated, tile by tile,
// so that a step needs no split launch of its own).  Plane format: see the header of ep_planes.hip.
#pragma once
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
// one row-major matrix W (R x K) and where its planes go
// ---------------------------------------------------------------------------------------------------------------------
struct PlaneJob {
  const float* W; int R, K; int64_t ldw;            // row-major R x K
  uint16_t* pn; int64_t pn_term, pn_ld;             // planes of W   : [3][R][pn_ld],  pn_ld = round_up(K, 32)
  uint16_t* pt; int64_t pt_term, pt_ld;             // planes of W^T : [3][K][pt_ld],  pt_ld = round_up(R, 32)
};

// The 64 x 64 tile of W at (r0, c0), held in `tile` (zero outside the matrix), -> its planes in both orientations
// (either may be null).  256 threads; the caller has synchronised behind the tile's stores.
__device__ __forceinline__ void pl_emit_tile(const PlaneJob& jb, const float (*tile)[65], int r0, int c0, int tid) {
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

}  // namespace ep
