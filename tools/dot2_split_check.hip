// Is v_dot2c_f32_bf16 usable for the residuals of the three-term bf16 split (x = h + m + l)?  r = x - fp32(h) as
// dot2c(h_pair, (-1, 0), x): exact when the instruction adds the two (exact) products to the accumulator without an
// intermediate rounding that matters and does not flush what the plain subtraction keeps.  Compares, bit for bit, with
// v_sub_f32 over 2^26 random values of many magnitudes.  hipcc --offload-arch=gfx950 -O3 tools/dot2_split_check.hip -o /tmp/d2 && /tmp/d2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_rne(float a, float b) { f2v v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2)); }

__global__ void check(unsigned long long* bad, unsigned seed, int emin, int emax, unsigned selA_bits, unsigned selB_bits, float* dbg) {
  unsigned s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
  unsigned long long nb = 0;
  for (int it = 0; it < 1024; ++it) {
    float v[2];
    for (int j = 0; j < 2; ++j) {
      s = s * 1664525u + 1013904223u; const unsigned man = s >> 9;
      s = s * 1664525u + 1013904223u; const int e = emin + (int)((s >> 8) % (unsigned)(emax - emin + 1));
      s = s * 1664525u + 1013904223u; const unsigned sg = s >> 31;
      v[j] = __uint_as_float((sg << 31) | ((unsigned)(e + 127) << 23) | man);
    }
    // reference
    const unsigned h = pack_rne(v[0], v[1]);
    const float r0 = v[0] - __uint_as_float(h << 16), r1 = v[1] - __uint_as_float(h & 0xffff0000u);
    const unsigned m = pack_rne(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    const unsigned l = pack_rne(s0, s1);
    // dot2c
    const bf2 selA = __builtin_bit_cast(bf2, selA_bits), selB = __builtin_bit_cast(bf2, selB_bits);   // (kernel arguments: no inline-constant encoding in play)
    const bf2 hb = __builtin_bit_cast(bf2, h);
    const float q0 = __builtin_amdgcn_fdot2_f32_bf16(hb, selA, v[0], false), q1 = __builtin_amdgcn_fdot2_f32_bf16(hb, selB, v[1], false);
    const unsigned m2 = pack_rne(q0, q1);
    const bf2 mb = __builtin_bit_cast(bf2, m2);
    const float t0 = __builtin_amdgcn_fdot2_f32_bf16(mb, selA, q0, false), t1 = __builtin_amdgcn_fdot2_f32_bf16(mb, selB, q1, false);
    const unsigned l2 = pack_rne(t0, t1);
    if (__float_as_uint(q0) != __float_as_uint(r0) || __float_as_uint(q1) != __float_as_uint(r1) || m2 != m || l2 != l) {
      if (nb == 0 && blockIdx.x == 0 && threadIdx.x == 0) { dbg[0] = v[0]; dbg[1] = v[1]; dbg[2] = r0; dbg[3] = r1; dbg[4] = q0; dbg[5] = q1; dbg[6] = __uint_as_float(h << 16); dbg[7] = __uint_as_float(h & 0xffff0000u); }
      ++nb;
    }
  }
  atomicAdd(bad, nb);
}

int main() {
  unsigned long long* bad; hipMalloc(&bad, 8);
  const int ranges[][2] = {{-10, 10}, {-40, 40}, {-100, -60}, {-126, -110}, {60, 100}, {0, 0}};
  float* dbg; hipMalloc(&dbg, 64);
  const unsigned sels[][2] = {{0x0000bf80u, 0xbf800000u}, {0xbf800000u, 0x0000bf80u}};
  for (auto& sl : sels)
  for (auto& r : ranges) {
    hipMemset(bad, 0, 8); hipMemset(dbg, 0, 64);
    hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, bad, 12345u, r[0], r[1], sl[0], sl[1], dbg);
    unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    float d[8]; hipMemcpy(d, dbg, 32, hipMemcpyDeviceToHost);
    printf("sel %08x/%08x exponents %4d .. %4d: %llu of %llu pairs differ from the v_sub_f32 split", sl[0], sl[1], r[0], r[1], h, 1024ull * 256 * 1024);
    if (h) printf("   e.g. x = (%a, %a) h = (%a, %a): sub (%a, %a) dot2c (%a, %a)", d[0], d[1], d[6], d[7], d[2], d[3], d[4], d[5]);
    printf("\n");
  }
  return 0;
}
