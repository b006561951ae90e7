// Ring-only ablation of the token passes (round 5): how fast can a persistent workgroup grid move
// B x N x D fp32 tokens HBM -> LDS through an LDS-DMA ring, as a function of the ring's geometry -- waves per workgroup,
// bytes per ring tile, ring depth, workgroups per CU -- with NO arithmetic at all (the consumer is one s_barrier per tile)?
// Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/ring_only.hip -o /tmp/ring_only && /tmp/ring_only [B N D]
// Prints one line per geometry: time per pass (median of 20, 3 token buffers rotated so nothing is served from the
// Infinity Cache), TB/s.  The geometry the vector-ALU passes use today is (4 waves, 12 KiB tiles, 3 workgroups per CU);
// the matrix-core passes use (8 waves, 48 KiB tiles, 1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ __forceinline__ void wait_vmcnt(int n) {
#define W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20)
    W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40)
    W(41) W(42) W(43) W(44) W(45) W(46) W(47) W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60)
    default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
  }
#undef W
}

// NW waves; a tile = NW * KPW pieces of 1 KiB (one global_load_lds of 16 B per lane each); nslot ring slots; `nbar` barriers per tile
template <int NW, int KPW>
__global__ __launch_bounds__(NW * 64) void ring_kernel(const char* x, long img_bytes, int B, int nslot, int nbar, int aux, float* sink, int rowb) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  constexpr int TILE = NW * KPW * 1024;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, wg = blockIdx.x;
  const int tiles_per_img = (int)((img_bytes + TILE - 1) / TILE);
  const int n_img = (B - wg + G - 1) / G;
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* psrc = x + (long)wg * img_bytes;
  auto produce = [&]() {
    if (pi < n_items) {
      const long left = img_bytes - (long)ptile * TILE;
      const unsigned limit = (unsigned)((left < TILE ? left : TILE) - 16);
#pragma unroll
      for (int k = 0; k < KPW; ++k) {
        unsigned off = (unsigned)((w + NW * k) * 1024 + lane * 16);
        if (rowb) {                                    // the passes' source-side swizzle: LDS chunk (t, c') <- source chunk c' ^ (t & 15) of row t
          const unsigned nch = (unsigned)rowb >> 4, pos = off >> 4, t = pos / nch, c = pos - t * nch;
          off = t * (unsigned)rowb + ((c ^ (t & 15u)) << 4);
        }
        off = off < limit ? off : limit;
        if (aux) __builtin_amdgcn_global_load_lds((gptr_t)(psrc + off), (lds_ptr_t)(ring + pslot * TILE + (w + NW * k) * 1024), 16, 0, 2);
        else __builtin_amdgcn_global_load_lds((gptr_t)(psrc + off), (lds_ptr_t)(ring + pslot * TILE + (w + NW * k) * 1024), 16, 0, 0);
      }
      ++pi;
      pslot = (pslot + 1 == nslot) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        const int bn = (wg + pimg * G) < B ? (wg + pimg * G) : wg;
        psrc = x + (long)bn * img_bytes;
      } else {
        psrc += TILE;
      }
    }
  };
  for (int s = 0; s < nslot - 1; ++s) produce();
  float acc = 0.f;
  int cslot = 0;
  for (int i = 0; i < n_items; ++i) {
    wait_vmcnt((pi - 1 - i) * KPW);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    produce();
    if (i == n_items - 1) acc += *reinterpret_cast<const float*>(ring + cslot * TILE + threadIdx.x * 4);   // (keeps the ring observable)
    cslot = (cslot + 1 == nslot) ? 0 : cslot + 1;
    for (int k = 1; k < nbar; ++k) asm volatile("s_barrier" ::: "memory");
  }
  if (acc == 123.456f) sink[0] = acc;
}

struct Geo { int nw, kpw, wgs, nslot, nbar, aux; };

template <int NW, int KPW>
static float run(const Geo& g, const std::vector<char*>& xs, long img_bytes, int B, int cus, float* sink, int rowb = 0, int lds_pad = 0) {
  const int tile = NW * KPW * 1024;
  const size_t lds = (size_t)g.nslot * tile + lds_pad;
  auto k = ring_kernel<NW, KPW>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1.f;
  int grid = cus * g.wgs; if (grid > B) grid = B;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 25; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, xs[it % xs.size()], img_bytes, B, g.nslot, g.nbar, g.aux, sink, rowb);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return -1.f;
    if (it >= 5) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 256, D = argc > 3 ? atoi(argv[3]) : 768;
  const long img_bytes = (long)N * D * 4;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<char*> xs(3);
  for (auto& p : xs) { hipMalloc(&p, (size_t)B * img_bytes); hipMemset(p, 1, (size_t)B * img_bytes); }
  float* sink; hipMalloc(&sink, 64);
  const double gb = (double)B * img_bytes / 1e9;
  printf("# ring-only token pass, %d x %d x %d fp32 = %.1f MB per pass, %d CUs; columns: waves/WG tile_KiB WGs/CU slots barriers/tile nt | us | TB/s\n", B, N, D, gb * 1e3, cus);
  const int LDS = 160 * 1024;
  for (int nt = 1; nt >= 0; --nt)
  for (int nw : {4, 8})
    for (int kpw : {1, 2, 3, 6, 12})
      for (int wgs : {1, 2, 3, 4}) {
        const int tile = nw * kpw * 1024;
        if (tile > 96 * 1024) continue;
        if (nw * wgs > 16) continue;
        const int maxslot = (LDS / wgs) / tile;
        int deepest = maxslot > 12 ? 12 : maxslot;
        while (deepest > 2 && (deepest - 1) * kpw > 60) --deepest;                        // vmcnt is a 6-bit counter
        for (int nslot = 2; nslot <= deepest; ++nslot) {
          if (nslot != deepest && nslot != 3 && nslot != 4 && nslot != 8) continue;       // (depth: the deepest that fits, and 3 / 4 / 8)
          if (nt == 0 && nslot != deepest) continue;                                      // default cache policy: only the deepest ring
          Geo g{nw, kpw, wgs, nslot, 1, nt};
          float us = -1.f;
#define RUN(NW_, KPW_) if (nw == NW_ && kpw == KPW_) us = run<NW_, KPW_>(g, xs, img_bytes, B, cus, sink);
          RUN(4, 1) RUN(4, 2) RUN(4, 3) RUN(4, 6) RUN(4, 12) RUN(8, 1) RUN(8, 2) RUN(8, 3) RUN(8, 6) RUN(8, 12)
#undef RUN
          if (us > 0) printf("%d %3d %d %2d %d %d | %7.1f | %.2f\n", nw, tile / 1024, wgs, nslot, 1, nt, us, gb / us * 1e3);
          fflush(stdout);
        }
      }
  // the matrix-core passes' geometry (8 waves, 48 KiB tiles, one workgroup per CU, 3 slots) with the source-side row swizzle of
  // the passes, for fp32 rows (D x 4 bytes) and bf16 rows (D x 2 bytes), one and two barriers per tile
  for (int rowb : {0, D * 4, D * 2})
    for (int nbar : {1, 2}) {
      Geo g{8, 6, 1, 3, nbar, 1};
      const float us = run<8, 6>(g, xs, img_bytes, B, cus, sink, rowb);
      printf("swizzled-source: 8 waves 48 KiB 1 WG/CU 3 slots, row bytes %d, %d barrier(s) per tile | %7.1f | %.2f\n", rowb, nbar, us, gb / us * 1e3);
      if (nbar == 2) {
        const float us2 = run<8, 6>(g, xs, img_bytes, B, cus, sink, rowb, 16384);
        printf("   ... with the whole 160 KiB of LDS allocated (ring + 16 KiB)                          | %7.1f | %.2f\n", us2, gb / us2 * 1e3);
      }
    }
  return 0;
}
