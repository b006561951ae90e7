// Shared host/device helpers for the EP HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ep_hip.h"

namespace ep {

void set_error(const char* fmt, ...);

#define EP_REQUIRE(cond, code, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      ep::set_error(__VA_ARGS__);          \
      return (code);                       \
    }                                      \
  } while (0)

#define EP_LAUNCH_CHECK(what)                                            \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      ep::set_error("%s: %s", (what), hipGetErrorString(e__));           \
      return (int)e__;                                                   \
    }                                                                    \
  } while (0)

#define EP_TRY(expr)          \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != 0) return rc__; \
  } while (0)

#define EP_HIP(expr)                                                     \
  do {                                                                   \
    hipError_t e__ = (expr);                                             \
    if (e__ != hipSuccess) {                                             \
      ep::set_error("%s: %s", #expr, hipGetErrorString(e__));            \
      return (int)e__;                                                   \
    }                                                                    \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int cu_count();

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---- bf16 token storage: four consecutive bf16 values (8 bytes) widened to fp32 (exact) --------
__device__ __forceinline__ f4 bf16x4_to_f4(uint2 v) {
  f4 r;
  r.x = __uint_as_float(v.x << 16); r.y = __uint_as_float(v.x & 0xffff0000u);
  r.z = __uint_as_float(v.y << 16); r.w = __uint_as_float(v.y & 0xffff0000u);
  return r;
}
// ---- fp16 token storage (forward / evaluation only, ABI v24): four consecutive fp16 values widened to fp32 (exact) ----
__device__ __forceinline__ f4 f16x4_to_f4(uint2 v) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const h2v a = __builtin_bit_cast(h2v, v.x), b = __builtin_bit_cast(h2v, v.y);
  return f4{(float)a.x, (float)a.y, (float)b.x, (float)b.y};
}
// 16-bit stored tokens: bf16, or (f16 != 0, a wave-uniform run-time flag of the 16-bit instantiations) fp16
__device__ __forceinline__ f4 h16x4_to_f4(uint2 v, int f16) { return f16 ? f16x4_to_f4(v) : bf16x4_to_f4(v); }
// tokens are fp32 or 16-bit (bf16 / fp16) in memory; all arithmetic is fp32.  `e`: element index (multiple of 4)
template <bool BF16>
__device__ __forceinline__ f4 load_tok4(const void* base, int64_t e, int f16 = 0) {
  if (BF16) return h16x4_to_f4(*reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + e), f16);
  return *reinterpret_cast<const f4*>(static_cast<const float*>(base) + e);
}

// ---- wave64 helpers -------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

// DPP row operations (16-lane rows).  ctrl: quad_perm [1,0,3,2]=0xB1, [2,3,0,1]=0x4E,
// row_half_mirror=0x141, row_mirror=0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(
      __builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// sum of the 16 lanes of each row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
// fold two registers across the 32-lane halves: result lanes 0-31 = sum over both halves of
// a, lanes 32-63 = sum over both halves of b  (v_permlane32_swap, gfx950).
__device__ __forceinline__ float fold32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// fold two registers across adjacent 16-lane rows: rows of the result hold
// [a0+a1, b0+b1, a2+a3, b2+b3]  (v_permlane16_swap, gfx950).
__device__ __forceinline__ float fold16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// full-wave sum, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v = fold16(v, v);   // rows: [r0+r1, r0+r1, r2+r3, r2+r3]
  v = fold32(v, v);   // all lanes: total
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace ep
