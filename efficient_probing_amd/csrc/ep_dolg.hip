// DOLG spatial attention pooling (reference poolings/dolg/dolg.py:11-62 SpatialAttention2d; registry entry
// probe_heads.py:82: SpatialAttention2d(in_c=dim, s3_dim=dim, with_aspp=False)).  On the token grid:
//     Y = conv1(x)            1x1 convolution = one (B N) x D x D contraction + bias            (dolg.py:47)
//     Yh = BatchNorm2d(Y)     batch statistics over all B N tokens per channel, affine, eps 1e-5 (dolg.py:48)
//     F = Yh / max(|Yh|_2, 1e-12)  per token ;  s = conv2(relu(Yh))  (D -> 1) ;  att = softplus(s)   (dolg.py:50-55)
//     out[b] = mean_n att[b,n] F[b,n]                                                              (dolg.py:57,60)
// This head is matrix-core bound like AbMILP: 2 N D^2 FLOP per image forward, the same again for d conv1.weight (the tokens
// are frozen: no gradient w.r.t. x).  The contraction runs on the exact-fp32 MFMA kernel (ep_gemm.hip), the BatchNorm over
// the token rows on the head's own two-stage BatchNorm kernels (ep_tail.hip, rows = B N), and everything between the
// BatchNorm and the pooled vector is ONE streaming kernel per direction over the normalised activations (a wave per token
// row, attention scalar and norm by wave reductions, the pooled vector in registers).
// d conv1.bias is exactly zero: the BatchNorm removes a per-channel shift of Y (the reference holds rounding noise).
#include "ep_side.h"

namespace ep {

__device__ __forceinline__ float dl_dot4(f4 a, f4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ f4 dl_relu(f4 v) { return f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }
__device__ __forceinline__ f4 dl_gate(f4 v, f4 y) {      // v where y > 0 (the ReLU's derivative)
  return f4{y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f};
}

// One workgroup (4 waves) per image; wave w takes tokens n = w, w + 4, ...; a lane holds CPL 16-byte chunks of the row.
//   forward : tok[b,n] = {att, 1 / max(|Yh|, 1e-12), s} ; out[b,:] = mean_n att F
//   backward: dZ[b,n,:] = gamma * dYh ; per-image partials part[b] = {sum dYh z, sum dYh, sum ds relu(Yh)} (3 D) and ds sum
template <int CPL, bool BWD>
__global__ __launch_bounds__(256) void ep_dolg_row_kernel(const float* __restrict__ Z, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ w2,
                                                        const float* __restrict__ c2p, int N, int D, float* __restrict__ tok,
                                                        float* __restrict__ out,
                                                        const float* __restrict__ dout, float* __restrict__ dZ,
                                                        float* __restrict__ part, float* __restrict__ dssum) {
  extern __shared__ __attribute__((aligned(16))) float dl_lds[];
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const float c2 = c2p[0];
  int ch[CPL];
  bool cv[CPL];
  f4 g[CPL], be[CPL], wv[CPL], dv[BWD ? CPL : 1];
  f4 a0[CPL], a1[BWD ? CPL : 1], a2[BWD ? CPL : 1];
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = 4 * (lane + 64 * j);
    cv[j] = c < D;
    ch[j] = cv[j] ? c : 0;
    g[j] = cv[j] ? *reinterpret_cast<const f4*>(gamma + ch[j]) : zero;
    be[j] = cv[j] ? *reinterpret_cast<const f4*>(beta + ch[j]) : zero;
    wv[j] = cv[j] ? *reinterpret_cast<const f4*>(w2 + ch[j]) : zero;
    a0[j] = zero;
    if constexpr (BWD) {
      dv[j] = cv[j] ? *reinterpret_cast<const f4*>(dout + (int64_t)b * D + ch[j]) * (1.0f / (float)N) : zero;
      a1[j] = zero; a2[j] = zero;
    }
  }
  float dsacc = 0.f;
  for (int n = w; n < N; n += 4) {
    const int64_t row = ((int64_t)b * N + n) * D;
    f4 y[CPL];
    f4 z[CPL];
    float nrm = 0.f, sd = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      z[j] = cv[j] ? *reinterpret_cast<const f4*>(Z + row + ch[j]) : zero;
      y[j] = cv[j] ? g[j] * z[j] + be[j] : zero;
      nrm += dl_dot4(y[j], y[j]);
      sd += dl_dot4(dl_relu(y[j]), wv[j]);
    }
    nrm = wave_sum(nrm);
    if constexpr (!BWD) {
      sd = wave_sum(sd) + c2;
      const float att = sd > 20.f ? sd : log1pf(expf(sd));            // nn.Softplus(beta=1, threshold=20)
      const float inv = 1.0f / fmaxf(sqrtf(nrm), 1e-12f);              // F.normalize(p=2, dim=1, eps=1e-12)
      const float wt = att * inv;
#pragma unroll
      for (int j = 0; j < CPL; ++j) a0[j] += wt * y[j];
      if (lane == 0) {
        float* t = tok + ((int64_t)b * N + n) * 4;
        t[0] = att; t[1] = inv; t[2] = sd;
      }
    } else {
      const float* t = tok + ((int64_t)b * N + n) * 4;
      const float att = t[0], inv = t[1], s = t[2];
      float gf = 0.f;                                                  // g . F
#pragma unroll
      for (int j = 0; j < CPL; ++j) gf += dl_dot4(dv[j], y[j]);
      gf = wave_sum(gf) * inv;
      const float ds = gf * (s > 20.f ? 1.0f : 1.0f / (1.0f + expf(-s)));   // d softplus = sigmoid
      dsacc += ds;
      const float ai = att * inv;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        // dYh = ds w2 [Yh > 0] + att inv (g - F (g . F))     with F = Yh inv
        const f4 dy = dl_gate(ds * wv[j], y[j]) + ai * (dv[j] - (gf * inv) * y[j]);
        if (cv[j]) *reinterpret_cast<f4*>(dZ + row + ch[j]) = dy * g[j];
        a0[j] += dy * z[j];
        a1[j] += dy;
        a2[j] += ds * dl_relu(y[j]);
      }
    }
  }
  // merge the four token waves (fixed order)
  constexpr int NV = BWD ? 3 : 1;
  float* rec = dl_lds + ((size_t)w * 64 + lane) * (4 * CPL * NV + 4);
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    *reinterpret_cast<f4*>(rec + 4 * j) = a0[j];
    if constexpr (BWD) {
      *reinterpret_cast<f4*>(rec + 4 * (CPL + j)) = a1[j];
      *reinterpret_cast<f4*>(rec + 4 * (2 * CPL + j)) = a2[j];
    }
  }
  rec[4 * CPL * NV] = dsacc;
  __syncthreads();
  if (w != 0) return;
  for (int i = 1; i < 4; ++i) {
    const float* o = dl_lds + ((size_t)i * 64 + lane) * (4 * CPL * NV + 4);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      a0[j] += *reinterpret_cast<const f4*>(o + 4 * j);
      if constexpr (BWD) {
        a1[j] += *reinterpret_cast<const f4*>(o + 4 * (CPL + j));
        a2[j] += *reinterpret_cast<const f4*>(o + 4 * (2 * CPL + j));
      }
    }
    dsacc += o[4 * CPL * NV];
  }
  if constexpr (!BWD) {
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (cv[j]) *reinterpret_cast<f4*>(out + (int64_t)b * D + ch[j]) = a0[j] * invN;
  } else {
    float* pb = part + (int64_t)b * 3 * D;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (cv[j]) {
        *reinterpret_cast<f4*>(pb + ch[j]) = a0[j];
        *reinterpret_cast<f4*>(pb + D + ch[j]) = a1[j];
        *reinterpret_cast<f4*>(pb + 2 * D + ch[j]) = a2[j];
      }
    if (lane == 0) dssum[b] = dsacc;                                // (every lane carries the same sums)
  }
}

// out[0] (+)= sum_b v[b]     (one workgroup, fixed order)
__global__ __launch_bounds__(256) void ep_dolg_scalar_sum_kernel(const float* __restrict__ v, int B, int accumulate,
                                                               float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += v[b];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = accumulate ? out[0] + t : t;
  }
}

// dst[i] += src[i]
__global__ void ep_dolg_scalar_add_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

template <bool BWD>
static int dolg_rows(int D, int B, int N, const float* Z, const float* gamma, const float* beta, const float* w2, const float* c2,
                     float* tok, float* out, const float* dout, float* dZ, float* part, float* dssum, hipStream_t st) {
  const int cpl = (D / 4 + 63) / 64;
  constexpr int NV = BWD ? 3 : 1;
#define EP_DL(C_)                                                                                                   \
  if (cpl <= C_) {                                                                                                  \
    const size_t lds = (size_t)4 * 64 * (4 * C_ * NV + 4) * sizeof(float);                                          \
    auto kf = ep_dolg_row_kernel<C_, BWD>;                                                                          \
    if (lds > 48 * 1024) {                                                                                          \
      hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; } \
    }                                                                                                               \
    hipLaunchKernelGGL(kf, dim3(B), dim3(256), lds, st, Z, gamma, beta, w2, c2, N, D, tok, out, dout, dZ, part, dssum); \
    EP_LAUNCH_CHECK("ep_dolg_row_kernel");                                                                          \
    return 0;                                                                                                       \
  }
  EP_DL(1) EP_DL(2) EP_DL(3) EP_DL(4) EP_DL(5) EP_DL(8)
#undef EP_DL
  set_error("dolg: D = %d too wide for the row kernel (D <= 2048)", D);
  return EP_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
constexpr int DOLG_NT = 8;    // conv1.weight conv1.bias | bn.weight bn.bias | conv2.weight conv2.bias | fc.weight fc.bias
constexpr int DOLG_KSPLIT = 8;
struct DolgWs {
  float *Z, *G, *rstd, *tok, *part, *dssum, *wpart, *bnpart;
  uint16_t* plC1;                  // planes of conv1.weight: the 1x1 convolution on the bf16-plane kernel (ep_planes.hip)
  float *y, *z, *hrstd, *logits, *dlogits, *rowstat, *hbnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static void dolg_sizes(const ep_dolg_dims& d, int64_t sizes[DOLG_NT]) {
  const int64_t D = d.D;
  const int64_t s[DOLG_NT] = {D * D, D, D, D, D, 1, (int64_t)d.C * D, d.C};
  for (int i = 0; i < DOLG_NT; ++i) sizes[i] = s[i];
}
static int64_t dolg_offsets(const ep_dolg_dims& d, int64_t offs[DOLG_NT]) {
  int64_t sizes[DOLG_NT];
  dolg_sizes(d, sizes);
  int64_t off = 0;
  for (int i = 0; i < DOLG_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static DolgWs dolg_carve(const ep_dolg_dims& d, void* base, bool head) {
  DolgWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, R = (size_t)d.B * d.N;
  w.Z = take(R * D); w.G = take(R * D); w.rstd = take(D); w.tok = take(R * 4);
  w.part = take((B + 16) * 3 * D + 16 * 3 * D); w.dssum = take(B);
  w.wpart = take((size_t)(DOLG_KSPLIT + 16) * D * D);
  w.bnpart = take(bn_workspace_bytes((int)R, d.D) / sizeof(float));
  {
    static int on = -1;            // EP_DOLG_PLANES=0: the convolution on the f32 matrix instruction (rounds 1 - 2)
    if (on < 0) { const char* e = getenv("EP_DOLG_PLANES"); on = e ? atoi(e) : 1; }
    if (on) w.plC1 = reinterpret_cast<uint16_t*>(take((planes_elems(d.D, d.D) + 1) / 2));
  }
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.hrstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.hbnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[DOLG_NT];
    w.opt_ws_bytes = optim_workspace_bytes(dolg_offsets(d, offs), DOLG_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int dolg_check(const ep_dolg_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0, EP_E_ARG, "dolg dims must be positive");
  EP_REQUIRE(d.D % 4 == 0 && d.D <= 2048, EP_E_SHAPE, "dolg: D must be a multiple of 4, at most 2048 (D=%d)", d.D);
  EP_REQUIRE((int64_t)d.B * d.N < (1ll << 31), EP_E_SHAPE, "dolg: B * N overflows");
  const int side = (int)lround(sqrt((double)d.N));
  EP_REQUIRE(side * side == d.N, EP_E_SHAPE, "dolg: N = %d is not a square token grid (the reference's view(b, c, h, w) fails too)", d.N);
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "dolg head: C must be positive");
  return 0;
}

static int dolg_params_ok(const ep_dolg_params* p, const char* what) {
  EP_REQUIRE(p, EP_E_ARG, "%s: null parameter struct", what);
  const float* ts[] = {p->conv1_w, p->conv1_b, p->bn_w, p->bn_b, p->conv2_w, p->conv2_b};
  for (const float* t : ts) EP_REQUIRE(t && aligned16(t), EP_E_ALIGN, "%s: tensors must be non-null and 16-byte aligned", what);
  return 0;
}

static GemmParams dg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

struct DolgBn { int training; float eps, momentum; float *running_mean, *running_var; int64_t* nbt; };

// x: (B*N, D) fp32, contiguous
static int dolg_forward_core(const ep_dolg_dims& d, const float* x, const DolgBn& bn, const ep_dolg_params& pr, const DolgWs& w,
                             float* y, hipStream_t st) {
  const int D = d.D, R = d.B * d.N;
  if (w.plC1) {                                                                                                          // Y = x W1^T + c1
    PlaneSpec sp{pr.conv1_w, D, D, D, w.plC1, nullptr};
    EP_TRY(planes_split(&sp, 1, st));
    GemmParams g{};
    g.A = x; g.lda = D; g.C = w.Z; g.ldc = D; g.M = R; g.N = D; g.K = D; g.alpha = 1.f; g.bias = pr.conv1_b;
    g.Bpl = w.plC1; g.ldbp = (int64_t)round_up((size_t)D, 32); g.pl_term = (int64_t)D * g.ldbp;
    EP_TRY(gemm_planes(g, 1, st));
  } else {
    GemmParams g = dg(x, D, pr.conv1_w, D, w.Z, D, R, D, D); g.bias = pr.conv1_b; EP_TRY(gemm(true, true, g, 1, st));
  }
  if (bn.training) {
    EP_TRY(bn_forward_train(w.Z, R, D, bn.eps, bn.momentum, w.Z, w.rstd, bn.running_mean, bn.running_var, bn.nbt, w.bnpart, st));
  } else {
    EP_REQUIRE(bn.running_mean && bn.running_var, EP_E_ARG, "dolg eval: running statistics missing");
    EP_TRY(bn_forward_eval(w.Z, R, D, bn.eps, bn.running_mean, bn.running_var, w.Z, st));
  }
  return dolg_rows<false>(D, d.B, d.N, w.Z, pr.bn_w, pr.bn_b, pr.conv2_w, pr.conv2_b, w.tok, y, nullptr, nullptr, nullptr, nullptr, st);
}

static int dolg_backward_core(const ep_dolg_dims& d, const float* x, const ep_dolg_params& pr, const float* dy,
                              const ep_dolg_params& gr, int acc, const DolgWs& w, hipStream_t st) {
  const int D = d.D, R = d.B * d.N, B = d.B;
  EP_TRY(dolg_rows<true>(D, B, d.N, w.Z, pr.bn_w, pr.bn_b, pr.conv2_w, pr.conv2_b, w.tok, nullptr, dy, w.G, w.part, w.dssum, st));
  // d bn.weight | d bn.bias | d conv2.weight from the per-image partials (fixed order), d conv2.bias from the ds sums
  float* red = w.part + (size_t)B * 3 * D;                            // 3 D reduced values, then the reducer's stage
  EP_TRY(reduce_partials(w.part, B, 3 * D, 1.0f, 0, red, red + 3 * D, st));
  auto put = [&](const float* src, float* dst) -> int {
    if (acc) {
      hipLaunchKernelGGL(ep_dolg_scalar_add_kernel, dim3((D + 255) / 256), dim3(256), 0, st, src, D, dst);
    } else {
      EP_HIP(hipMemcpyAsync(dst, src, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return 0;
  };
  EP_TRY(put(red, gr.bn_w)); EP_TRY(put(red + D, gr.bn_b)); EP_TRY(put(red + 2 * D, gr.conv2_w));
  hipLaunchKernelGGL(ep_dolg_scalar_sum_kernel, dim3(1), dim3(256), 0, st, w.dssum, B, acc, gr.conv2_b);
  EP_LAUNCH_CHECK("ep_dolg gradient reductions");
  // BatchNorm backward over the B N rows (in place), then d conv1.weight = dY^T x in DOLG_KSPLIT deterministic K slices
  EP_TRY(bn_backward(w.G, w.Z, w.rstd, R, D, w.G, w.bnpart, st));
  int ks = DOLG_KSPLIT;
  while (ks > 1 && (R % ks != 0 || (R / ks) % 4 != 0)) ks >>= 1;
  {
    GemmParams g = dg(w.G, D, x, D, w.wpart, D, D, D, R / ks);
    g.sAz = (int64_t)(R / ks) * D; g.sBz = (int64_t)(R / ks) * D; g.sCz = (int64_t)D * D;
    EP_TRY(gemm(false, false, g, ks, st));
  }
  EP_TRY(reduce_partials(w.wpart, ks, D * D, 1.0f, acc, gr.conv1_w, w.wpart + (size_t)ks * D * D, st));
  if (!acc) EP_HIP(hipMemsetAsync(gr.conv1_b, 0, (size_t)D * sizeof(float), st));      // exactly zero (see the header)
  return 0;
}

static ep_dolg_params dolg_views(float* base, const int64_t o[DOLG_NT]) {
  ep_dolg_params p;
  p.conv1_w = base + o[0]; p.conv1_b = base + o[1]; p.bn_w = base + o[2]; p.bn_b = base + o[3]; p.conv2_w = base + o[4];
  p.conv2_b = base + o[5];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

static int dolg_tokens_ok(const ep_dolg_dims& d, const void* x, int x_dtype, int64_t x_bstride) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, 1));
  EP_REQUIRE(x_dtype == EP_DTYPE_F32 && x_bstride == (int64_t)d.N * d.D, EP_E_UNSUPPORTED,
             "dolg: the tokens feed a matrix-core contraction: a dense fp32 (B, N, D) tensor is required");
  return 0;
}

size_t ep_dolg_pool_workspace_bytes(const ep_dolg_dims* dims) {
  if (!dims || dolg_check(*dims, false) != 0) return 0;
  return dolg_carve(*dims, nullptr, false).total;
}

int ep_dolg_pool_forward(const ep_dolg_dims* dims, const void* x, int x_dtype, int64_t x_bstride, int training, float bn_eps,
                         float bn_momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         const ep_dolg_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_dolg_pool_forward: null pointer");
  EP_TRY(dolg_check(*dims, false));
  EP_TRY(dolg_params_ok(params, "ep_dolg_pool_forward"));
  EP_TRY(dolg_tokens_ok(*dims, x, x_dtype, x_bstride));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_dolg_pool_forward: y / ws must be 16-byte aligned");
  const DolgWs w = dolg_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dolg_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  const DolgBn bn{training, bn_eps, bn_momentum, running_mean, running_var, num_batches_tracked};
  return dolg_forward_core(*dims, static_cast<const float*>(x), bn, *params, w, y, (hipStream_t)stream);
}

int ep_dolg_pool_backward(const ep_dolg_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const ep_dolg_params* params,
                          const float* dy, const ep_dolg_params* grads, int accumulate, void* ws, size_t ws_bytes,
                          ep_stream_t stream) {
  EP_REQUIRE(dims && dy && ws, EP_E_ARG, "ep_dolg_pool_backward: null pointer");
  EP_TRY(dolg_check(*dims, false));
  EP_TRY(dolg_params_ok(params, "ep_dolg_pool_backward(params)"));
  EP_TRY(dolg_params_ok(grads, "ep_dolg_pool_backward(grads)"));
  EP_TRY(dolg_tokens_ok(*dims, x, x_dtype, x_bstride));
  EP_REQUIRE(aligned16(ws) && aligned16(dy), EP_E_ALIGN, "ep_dolg_pool_backward: dy / ws must be 16-byte aligned");
  const DolgWs w = dolg_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dolg_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return dolg_backward_core(*dims, static_cast<const float*>(x), *params, dy, *grads, accumulate, w, (hipStream_t)stream);
}

/* attention scores softplus(conv2(relu(bn(conv1 x)))) (B, N) of the last forward on this workspace (dolg.py:59 return_attn) */
int ep_dolg_attention(const ep_dolg_dims* dims, const void* ws, float* att, ep_stream_t stream) {
  EP_REQUIRE(dims && ws && att, EP_E_ARG, "ep_dolg_attention: null pointer");
  EP_TRY(dolg_check(*dims, false));
  const DolgWs w = dolg_carve(*dims, const_cast<void*>(ws), false);
  EP_HIP(hipMemcpy2DAsync(att, sizeof(float), w.tok, 4 * sizeof(float), sizeof(float), (size_t)dims->B * dims->N,
                          hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int64_t ep_dolg_head_param_offsets(const ep_dolg_dims* dims, int64_t offsets[8]) { return dolg_offsets(*dims, offsets); }

size_t ep_dolg_head_workspace_bytes(const ep_dolg_dims* dims) {
  if (!dims || dolg_check(*dims, true) != 0) return 0;
  return dolg_carve(*dims, nullptr, true).total;
}

int ep_dolg_head_train_step(const ep_dolg_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_dolg_head_train_step: null pointer");
  const ep_dolg_dims& d = s->dims;
  EP_TRY(dolg_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const DolgWs w = dolg_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dolg_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[DOLG_NT];
  const int64_t total = dolg_offsets(d, offs);
  const ep_dolg_params pr = dolg_views(s->params, offs), gr = dolg_views(s->grads, offs);
  float* Wc = s->params + offs[6]; float* bc = s->params + offs[7];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats && s->tok_running_mean && s->tok_running_var,
               EP_E_ARG, "train step: null input");
    EP_TRY(dolg_tokens_ok(d, s->x, s->x_dtype, s->x_bstride));
    const float* x = static_cast<const float*>(s->x);
    const DolgBn bn{1, s->tok_bn_eps, s->tok_bn_momentum, s->tok_running_mean, s->tok_running_var, s->tok_num_batches_tracked};
    EP_TRY(dolg_forward_core(d, x, bn, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.hrstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.hbnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, s->grads + offs[6], s->grads + offs[7],
                           s->accumulate, st));
    EP_TRY(bn_backward(w.dz, w.z, w.hrstd, d.B, d.D, w.dy, w.hbnpart, st));
    EP_TRY(dolg_backward_core(d, x, pr, w.dy, gr, s->accumulate, w, st));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    int64_t sizes[DOLG_NT];
    dolg_sizes(d, sizes);
    const int trust[DOLG_NT] = {1, 0, 0, 0, 1, 0, 1, 0};          // util/lars.py:22: ndim > 1 (the conv weights are 4-D)
    ep_segment segs[DOLG_NT];
    for (int i = 0; i < DOLG_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? DOLG_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_dolg_head_eval_forward(const ep_dolg_dims* dims, const void* x, int x_dtype, int64_t x_bstride, float tok_bn_eps,
                              const float* tok_running_mean, const float* tok_running_var, const float* params,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                              void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && tok_running_mean && tok_running_var && logits && ws, EP_E_ARG,
             "ep_dolg_head_eval_forward: null pointer");
  const ep_dolg_dims& d = *dims;
  EP_TRY(dolg_check(d, true));
  EP_TRY(dolg_tokens_ok(d, x, x_dtype, x_bstride));
  const DolgWs w = dolg_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_dolg_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[DOLG_NT];
  dolg_offsets(d, offs);
  const ep_dolg_params pr = dolg_views(const_cast<float*>(params), offs);
  const DolgBn bn{0, tok_bn_eps, 0.f, const_cast<float*>(tok_running_mean), const_cast<float*>(tok_running_var), nullptr};
  EP_TRY(dolg_forward_core(d, static_cast<const float*>(x), bn, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[6], params + offs[7], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
