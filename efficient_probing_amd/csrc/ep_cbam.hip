// CBAM pooling (reference poolings/cbam.py:104-139 CbamPooling with ChannelAttn :19-36 and SpatialAttn :57-68; registry entry
// probe_heads.py:77: CbamPooling(channels=dim, spatial_kernel_size=7)).  On the h x w token grid:
//     gc[b,c] = sigmoid(fc2 relu(fc1 avg_n x) + fc2 relu(fc1 max_n x))                       channel gate   (cbam.py:33-36)
//     x1 = x gc ;  gs[b,n] = sigmoid(BatchNorm2d(conv7x7([mean_c x1 ; max_c x1])))            spatial gate   (cbam.py:65-68)
//     out[b,c] = mean_n relu(x1 gs + x)                                                       (cbam.py:131-138)
// Both gates are in (0, 1), so relu(x (1 + gc gs)) = (1 + gc gs) relu(x):
//     out[b,c] = R0[b,c] + gc[b,c] T[b,c] ,   R0 = mean_n relu(x) ,   T[b,c] = mean_n gs[b,n] relu(x[b,n,c])
// HBM-bound: the head is a sequence of streaming passes over the frozen tokens with tiny dense work in between --
//   A  per image  avg, max, R0 per channel          (depends on the tokens only: a resident store computes the table once)
//   B  per token  mean_c / max_c (+ arg max) of x gc
//   C  per image  T = mean_n gs relu(x)
//   D  per token  d gs = (1/N) sum_c dT relu(x)                                     (backward)
//   E  per image  d gc += sum_n x (dm1 / D + dm2 [c = arg max])                     (backward)
// and a 7x7 convolution + one-channel BatchNorm2d on the (B, 2, h, w) maps.  The max over channels back-propagates to its
// first maximal channel (exact ties, which torch splits evenly, do not occur on real token data).
#include "ep_side.h"

namespace ep {

// ---- pass A: table[b] = {avg, max, R0} over the N tokens, per channel ----------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void ep_cbam_chan_kernel(const void* __restrict__ x, int64_t bstride, const int* __restrict__ index,
                                                         int N, int D, float* __restrict__ tab) {
  const int b = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= D) return;
  const int64_t e0 = (int64_t)(index ? index[b] : b) * bstride + c;
  f4 s = {0.f, 0.f, 0.f, 0.f}, r = s, m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll 8
  for (int n = 0; n < N; ++n) {
    const f4 v = load_tok4<BF16>(x, e0 + (int64_t)n * D);
    s += v;
    m = f4{fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w)};
    r += f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
  }
  const float inv = 1.0f / (float)N;
  float* o = tab + (int64_t)b * 3 * D + c;
  *reinterpret_cast<f4*>(o) = s * inv;
  *reinterpret_cast<f4*>(o + D) = m;
  *reinterpret_cast<f4*>(o + 2 * D) = r * inv;
}

// out[i] = table[index[i / rowlen] * rowlen + i % rowlen]      (rows of a cached per-image table)
__global__ void ep_cbam_gather_kernel(const float* __restrict__ table, const int* __restrict__ index, int64_t total, int rowlen,
                                      float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = table[(int64_t)index[i / rowlen] * rowlen + (i % rowlen)];
}

// ---- token-row kernels (one workgroup per image, wave w takes tokens w, w + 4, ...; a lane holds CPL chunks) --------
//   MODE 0 (pass B): maps[b,0,n] = mean_c x gc ; maps[b,1,n] = max_c x gc ; arg[b,n] = first maximal channel
//   MODE 1 (pass D): dgs[b,n] = (1/N) sum_c dT[b,c] relu(x[b,n,c])
template <int CPL, bool BF16, int MODE>
__global__ __launch_bounds__(256) void ep_cbam_row_kernel(const void* __restrict__ x, int64_t bstride, const int* __restrict__ index,
                                                        int N, int D, const float* __restrict__ vec, float* __restrict__ maps,
                                                        int* __restrict__ arg) {
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int64_t e0 = (int64_t)(index ? index[b] : b) * bstride;
  int ch[CPL];
  bool cv[CPL];
  f4 g[CPL];
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = 4 * (lane + 64 * j);
    cv[j] = c < D;
    ch[j] = cv[j] ? c : 0;
    g[j] = cv[j] ? *reinterpret_cast<const f4*>(vec + (int64_t)b * D + ch[j]) : zero;
  }
  for (int n = w; n < N; n += 4) {
    f4 v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[j] = load_tok4<BF16>(x, e0 + (int64_t)n * D + ch[j]);
    if (MODE == 0) {
      float s = 0.f, mx = -INFINITY;
      int am = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const f4 p = v[j] * g[j];
        if (cv[j]) {
          s += (p.x + p.y) + (p.z + p.w);
          if (p.x > mx) { mx = p.x; am = ch[j]; }
          if (p.y > mx) { mx = p.y; am = ch[j] + 1; }
          if (p.z > mx) { mx = p.z; am = ch[j] + 2; }
          if (p.w > mx) { mx = p.w; am = ch[j] + 3; }
        }
      }
      s = wave_sum(s);
      const float wm = wave_max(mx);
      // first maximal channel: the smallest channel index among the lanes that hold the maximum
      int cand = (mx == wm) ? am : 0x7fffffff;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_xor(cand, off, 64); cand = o < cand ? o : cand; }
      if (lane == 0) {
        maps[((int64_t)b * 2 + 0) * N + n] = s / (float)D;
        maps[((int64_t)b * 2 + 1) * N + n] = wm;
        arg[(int64_t)b * N + n] = cand;
      }
    } else {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const f4 r = f4{fmaxf(v[j].x, 0.f), fmaxf(v[j].y, 0.f), fmaxf(v[j].z, 0.f), fmaxf(v[j].w, 0.f)};
        s = fmaf(r.x, g[j].x, fmaf(r.y, g[j].y, fmaf(r.z, g[j].z, fmaf(r.w, g[j].w, s))));
      }
      s = wave_sum(s);
      if (lane == 0) maps[(int64_t)b * N + n] = s / (float)N;
    }
  }
}

// ---- per-image channel accumulations over the tokens (a thread owns 4 channels) --------------------------------------
//   MODE 0 (pass C): out[b,c] = R0 + gc T ,  T[b,c] = (1/N) sum_n gs[b,n] relu(x[b,n,c])   (T is stored too)
//   MODE 1 (pass E): E[b,c] = sum_n x[b,n,c] (dm1[b,n] / D + dm2[b,n] [c == arg[b,n]])
template <bool BF16, int MODE>
__global__ __launch_bounds__(256) void ep_cbam_acc_kernel(const void* __restrict__ x, int64_t bstride, const int* __restrict__ index,
                                                        int N, int D, const float* __restrict__ tw, const int* __restrict__ arg,
                                                        const float* __restrict__ tab, const float* __restrict__ gc,
                                                        float* __restrict__ T, float* __restrict__ out) {
  const int b = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= D) return;
  const int64_t e0 = (int64_t)(index ? index[b] : b) * bstride + c;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    const float* gs = tw + (int64_t)b * N;
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
      const f4 v = load_tok4<BF16>(x, e0 + (int64_t)n * D);
      s += gs[n] * f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    }
    s = s * (1.0f / (float)N);
    *reinterpret_cast<f4*>(T + (int64_t)b * D + c) = s;
    const f4 r0 = *reinterpret_cast<const f4*>(tab + (int64_t)b * 3 * D + 2 * D + c);
    const f4 g = *reinterpret_cast<const f4*>(gc + (int64_t)b * D + c);
    *reinterpret_cast<f4*>(out + (int64_t)b * D + c) = r0 + g * s;
  } else {
    const float* d1 = tw + (int64_t)b * 2 * N;
    const float* d2 = d1 + N;
    const int* am = arg + (int64_t)b * N;
    const float invD = 1.0f / (float)D;
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
      const f4 v = load_tok4<BF16>(x, e0 + (int64_t)n * D);
      const float a = d1[n] * invD, m2 = d2[n];
      const int k = am[n] - c;
      s += v * f4{a + (k == 0 ? m2 : 0.f), a + (k == 1 ? m2 : 0.f), a + (k == 2 ? m2 : 0.f), a + (k == 3 ? m2 : 0.f)};
    }
    *reinterpret_cast<f4*>(out + (int64_t)b * D + c) = s;
  }
}

// ---- 7x7 convolution on the (B, 2, h, w) maps, padding 3, no bias: conv[b,p] = sum_{ch,dy,dx} w[ch,dy,dx] maps[b,ch,p+d] ----
__global__ __launch_bounds__(256) void ep_cbam_conv_kernel(const float* __restrict__ maps, const float* __restrict__ wgt, int side,
                                                         int KS, int64_t total, float* __restrict__ conv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int N = side * side, pad = KS / 2;
  const int64_t b = i / N; const int p = (int)(i % N), py = p / side, px = p % side;
  float acc = 0.f;
  for (int chn = 0; chn < 2; ++chn)
    for (int dy = 0; dy < KS; ++dy) {
      const int y = py + dy - pad;
      if (y < 0 || y >= side) continue;
      for (int dx = 0; dx < KS; ++dx) {
        const int xq = px + dx - pad;
        if (xq < 0 || xq >= side) continue;
        acc = fmaf(wgt[(chn * KS + dy) * KS + dx], maps[(b * 2 + chn) * N + y * side + xq], acc);
      }
    }
  conv[i] = acc;
}
// transposed: dmaps[b,ch,q] = sum_{dy,dx} w[ch,dy,dx] dconv[b, q - d]
__global__ __launch_bounds__(256) void ep_cbam_conv_bwd_kernel(const float* __restrict__ dconv, const float* __restrict__ wgt, int side,
                                                             int KS, int64_t total, float* __restrict__ dmaps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;                              // total = B * 2 * N
  const int N = side * side, pad = KS / 2;
  const int64_t bc = i / N; const int q = (int)(i % N), qy = q / side, qx = q % side;
  const int chn = (int)(bc % 2); const int64_t b = bc / 2;
  float acc = 0.f;
  for (int dy = 0; dy < KS; ++dy) {
    const int y = qy - dy + pad;
    if (y < 0 || y >= side) continue;
    for (int dx = 0; dx < KS; ++dx) {
      const int xq = qx - dx + pad;
      if (xq < 0 || xq >= side) continue;
      acc = fmaf(wgt[(chn * KS + dy) * KS + dx], dconv[b * N + y * side + xq], acc);
    }
  }
  dmaps[i] = acc;
}
// d w[ch,dy,dx] (+)= sum_{b,p} dconv[b,p] maps[b,ch,p+d]      (one workgroup per weight element, fixed order)
// grid (2 KS KS, RS): the images split over RS chunks -> part[chunk][2 KS KS]; ep_cbam_conv_wfin_kernel sums the chunks in order
__global__ __launch_bounds__(256) void ep_cbam_conv_wgrad_kernel(const float* __restrict__ dconv, const float* __restrict__ maps,
                                                               int B, int side, int KS, float* __restrict__ part) {
  __shared__ float red[4];
  const int e = blockIdx.x, chn = e / (KS * KS), dy = (e / KS) % KS, dx = e % KS;
  const int N = side * side, pad = KS / 2;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = (b0 + per) < B ? (b0 + per) : B;
  float s = 0.f;
  for (int64_t i = (int64_t)b0 * N + threadIdx.x; i < (int64_t)b1 * N; i += 256) {
    const int64_t b = i / N; const int p = (int)(i % N), y = p / side + dy - pad, xq = p % side + dx - pad;
    if (y >= 0 && y < side && xq >= 0 && xq < side) s = fmaf(dconv[i], maps[(b * 2 + chn) * N + y * side + xq], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)blockIdx.y * gridDim.x + e] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void ep_cbam_conv_wfin_kernel(const float* __restrict__ part, int n, int rs, int accumulate, float* __restrict__ dw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float t = 0.f;
  for (int r = 0; r < rs; ++r) t += part[(int64_t)r * n + e];
  dw[e] = accumulate ? dw[e] + t : t;
}

// ---- one-channel BatchNorm2d over all B*N values + sigmoid (single workgroup of 1024 threads, two passes, fixed order) ----
//   train: batch mean / biased variance, running statistics updated (unbiased variance) ; st = {mean, rstd}
// One-channel BatchNorm2d over all n = B N values + the sigmoid gate, on up to 64 workgroups:
//   ep_cbam_bn1_part_kernel : per chunk {count, mean, M2} (two passes inside the chunk)              [training only]
//   ep_cbam_bn1_kernel      : every workgroup combines the chunks in the same fixed order (Chan), applies its own chunk;
//                             workgroup 0 updates the running statistics and leaves {mean, rstd} in st
constexpr int CBAM_BN_G = 64;
__device__ __forceinline__ float cbam_block_sum(float v, float* red) {      // 1024 threads; result in every thread
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}
__device__ __forceinline__ void cbam_chunk(int64_t n, int64_t& c0, int64_t& c1) {
  const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 3) / 4 * 4;
  c0 = (int64_t)blockIdx.x * per; c0 = c0 < n ? c0 : n;
  c1 = (c0 + per) < n ? (c0 + per) : n;
}
__global__ __launch_bounds__(1024) void ep_cbam_bn1_part_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ part) {
  __shared__ float red[16];
  int64_t c0, c1;
  cbam_chunk(n, c0, c1);
  float s = 0.f;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 1024) s += v[i];
  const float cnt = (float)(c1 - c0);
  const float mean = cnt > 0.f ? cbam_block_sum(s, red) / cnt : 0.f;
  float q = 0.f;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 1024) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  q = cbam_block_sum(q, red);
  if (threadIdx.x == 0) { part[blockIdx.x * 4 + 0] = cnt; part[blockIdx.x * 4 + 1] = mean; part[blockIdx.x * 4 + 2] = q; }
}
__global__ __launch_bounds__(1024) void ep_cbam_bn1_kernel(const float* __restrict__ v, int64_t n, int training, float eps,
                                                         float momentum, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                                         const float* __restrict__ part, float* __restrict__ st,
                                                         float* __restrict__ gs) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    if (training) {
      float cn = 0.f, mu = 0.f, m2 = 0.f;                  // Chan's combination, chunks in order
      for (int g = 0; g < (int)gridDim.x; ++g) {
        const float nb = part[g * 4], mb = part[g * 4 + 1], qb = part[g * 4 + 2];
        if (nb > 0.f) {
          const float tot = cn + nb, dl = mb - mu;
          mu += dl * (nb / tot);
          m2 += qb + dl * dl * (cn * nb / tot);
          cn = tot;
        }
      }
      const float var = m2 / (float)n;
      bc[0] = mu; bc[1] = 1.0f / sqrtf(var + eps);
      if (blockIdx.x == 0) {
        if (rmean) {
          rmean[0] = (1.0f - momentum) * rmean[0] + momentum * mu;
          rvar[0] = (1.0f - momentum) * rvar[0] + momentum * (n > 1 ? m2 / (float)(n - 1) : var);
        }
        if (nbt) *nbt += 1;
      }
    } else {
      bc[0] = rmean[0]; bc[1] = 1.0f / sqrtf(rvar[0] + eps);
    }
    if (blockIdx.x == 0) { st[0] = bc[0]; st[1] = bc[1]; }
  }
  __syncthreads();
  const float mean = bc[0], rstd = bc[1];
  const float g = gamma[0], be = beta[0];
  int64_t c0, c1;
  cbam_chunk(n, c0, c1);
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 1024) gs[i] = 1.0f / (1.0f + expf(-fmaf(g, (v[i] - mean) * rstd, be)));
}
// backward: dpre = dgs gs (1 - gs) ; d gamma (+)= sum dpre zhat ; d beta (+)= sum dpre ;
//           dconv = gamma rstd (dpre - mean(dpre) - zhat mean(dpre zhat))          (batch statistics)
// backward of the gate + BatchNorm: ep_cbam_bn1_bwd_part_kernel leaves per chunk {sum dp, sum dp vhat}; ep_cbam_bn1_bwd_kernel
// sums the chunks in order in every workgroup and back-propagates its own chunk (workgroup 0 writes d gamma / d beta)
__global__ __launch_bounds__(1024) void ep_cbam_bn1_bwd_part_kernel(const float* __restrict__ dgs, const float* __restrict__ gs,
                                                                  const float* __restrict__ v, int64_t n,
                                                                  const float* __restrict__ st, float* __restrict__ part) {
  __shared__ float red[16];
  const float mean = st[0], rstd = st[1];
  int64_t c0, c1;
  cbam_chunk(n, c0, c1);
  float s1 = 0.f, s2 = 0.f;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 1024) {
    const float a = gs[i], dp = dgs[i] * a * (1.0f - a);
    s1 += dp; s2 = fmaf(dp, (v[i] - mean) * rstd, s2);
  }
  s1 = cbam_block_sum(s1, red);
  s2 = cbam_block_sum(s2, red);
  if (threadIdx.x == 0) { part[blockIdx.x * 4 + 0] = s1; part[blockIdx.x * 4 + 1] = s2; }
}
__global__ __launch_bounds__(1024) void ep_cbam_bn1_bwd_kernel(const float* __restrict__ dgs, const float* __restrict__ gs,
                                                             const float* __restrict__ v, int64_t n, const float* __restrict__ st,
                                                             const float* __restrict__ gamma, int accumulate,
                                                             const float* __restrict__ part, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dconv) {
  __shared__ float bc[2];
  const float mean = st[0], rstd = st[1];
  if (threadIdx.x == 0) {
    float t1 = 0.f, t2 = 0.f;
    for (int g = 0; g < (int)gridDim.x; ++g) { t1 += part[g * 4]; t2 += part[g * 4 + 1]; }
    if (blockIdx.x == 0) {
      dgamma[0] = accumulate ? dgamma[0] + t2 : t2;
      dbeta[0] = accumulate ? dbeta[0] + t1 : t1;
    }
    bc[0] = t1 / (float)n; bc[1] = t2 / (float)n;
  }
  __syncthreads();
  const float m1 = bc[0], m2 = bc[1], gr = gamma[0] * rstd;
  int64_t c0, c1;
  cbam_chunk(n, c0, c1);
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 1024) {
    const float a = gs[i], dp = dgs[i] * a * (1.0f - a);
    dconv[i] = gr * (dp - m1 - (v[i] - mean) * rstd * m2);
  }
}

// ---- channel gate: small element-wise kernels around the (B x D x rd) contractions ---------------------------------
// hs = relu(ha) + relu(hm)      (B, rd)
__global__ void ep_cbam_hsum_kernel(const float* __restrict__ ha, const float* __restrict__ hm, int64_t n, float* __restrict__ hs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hs[i] = fmaxf(ha[i], 0.f) + fmaxf(hm[i], 0.f);
}
__global__ void ep_cbam_sigmoid_kernel(const float* __restrict__ pre, int64_t n, float* __restrict__ g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] = 1.0f / (1.0f + expf(-pre[i]));
}
// dpre = (dout T + E) gc (1 - gc)       (B, D)
__global__ void ep_cbam_dpre_kernel(const float* __restrict__ dout, const float* __restrict__ T, const float* __restrict__ E,
                                    const float* __restrict__ gc, int64_t n, float* __restrict__ dpre, float* __restrict__ dT) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = gc[i];
  if (dT) dT[i] = dout[i] * g;
  if (dpre) dpre[i] = fmaf(dout[i], T[i], E[i]) * g * (1.0f - g);
}
// dha = dh [ha > 0] ; dhm = dh [hm > 0]
__global__ void ep_cbam_dh_kernel(const float* __restrict__ dh, const float* __restrict__ ha, const float* __restrict__ hm, int64_t n,
                                  float* __restrict__ dha, float* __restrict__ dhm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dha[i] = ha[i] > 0.f ? dh[i] : 0.f;
  dhm[i] = hm[i] > 0.f ? dh[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------
constexpr int CBAM_NT = 7;    // fc1.weight (rd,D) | fc2.weight (D,rd) | conv.weight (2 KS KS) | bn.weight | bn.bias | fc.weight fc.bias
constexpr int CBAM_WRS = 64;
static int cbam_bn_groups(int64_t n) { const int64_t g = n / 2048; return g < 1 ? 1 : (g > CBAM_BN_G ? CBAM_BN_G : (int)g); }
struct CbamWs {
  float *tab, *ha, *hm, *hs, *pre, *gc, *maps, *conv, *gs, *st, *T, *red, *wpart;
  int* arg;
  float *dT, *dgs, *dconv, *dmaps, *E, *dpre, *dh, *dha, *dhm;
  float *y, *z, *rstd, *logits, *dlogits, *rowstat, *bnpart, *dz, *dy;
  void* opt_ws; size_t opt_ws_bytes;
  int ldl;
  size_t total;
};

static void cbam_sizes(const ep_cbam_dims& d, int64_t sizes[CBAM_NT]) {
  const int64_t D = d.D, rd = d.rd, kk = (int64_t)2 * d.ks * d.ks;
  const int64_t s[CBAM_NT] = {rd * D, D * rd, kk, 1, 1, (int64_t)d.C * D, d.C};
  for (int i = 0; i < CBAM_NT; ++i) sizes[i] = s[i];
}
static int64_t cbam_offsets(const ep_cbam_dims& d, int64_t offs[CBAM_NT]) {
  int64_t sizes[CBAM_NT];
  cbam_sizes(d, sizes);
  int64_t off = 0;
  for (int i = 0; i < CBAM_NT; ++i) { offs[i] = off; off += (sizes[i] + 3) / 4 * 4; }
  return off;
}

static CbamWs cbam_carve(const ep_cbam_dims& d, void* base, bool head) {
  CbamWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up(nfloat * sizeof(float), 256);
    return p;
  };
  const size_t B = d.B, D = d.D, N = d.N, rd = d.rd;
  w.tab = take(B * 3 * D); w.ha = take(B * rd); w.hm = take(B * rd); w.hs = take(B * rd); w.pre = take(B * D); w.gc = take(B * D);
  w.maps = take(B * 2 * N); w.conv = take(B * N); w.gs = take(B * N); w.st = take(4); w.T = take(B * D); w.red = take(CBAM_BN_G * 4); w.wpart = take((size_t)CBAM_WRS * 2 * d.ks * d.ks);
  w.arg = reinterpret_cast<int*>(take(B * N));
  w.dT = take(B * D); w.dgs = take(B * N); w.dconv = take(B * N); w.dmaps = take(B * 2 * N); w.E = take(B * D); w.dpre = take(B * D);
  w.dh = take(B * rd); w.dha = take(B * rd); w.dhm = take(B * rd);
  if (head) {
    w.ldl = (d.C + 3) / 4 * 4;
    w.y = take(B * D); w.z = take(B * D); w.rstd = take(D);
    w.logits = take(B * w.ldl); w.dlogits = take(B * w.ldl); w.rowstat = take(B * 4);
    w.bnpart = take(bn_workspace_bytes(d.B, d.D) / sizeof(float));
    w.dz = take(B * D); w.dy = take(B * D);
    int64_t offs[CBAM_NT];
    w.opt_ws_bytes = optim_workspace_bytes(cbam_offsets(d, offs), CBAM_NT);
    w.opt_ws = take(w.opt_ws_bytes / sizeof(float));
  }
  w.total = off;
  return w;
}

static int cbam_side(int N) { const int s = (int)lround(sqrt((double)N)); return s * s == N ? s : 0; }

static int cbam_check(const ep_cbam_dims& d, bool head) {
  EP_REQUIRE(d.B > 0 && d.N > 0 && d.D > 0 && d.rd > 0 && d.ks > 0, EP_E_ARG, "cbam dims must be positive");
  EP_REQUIRE(d.D % 4 == 0 && d.D <= 4096, EP_E_SHAPE, "cbam: D must be a multiple of 4, at most 4096 (D=%d)", d.D);
  EP_REQUIRE(d.ks % 2 == 1 && d.ks <= 15, EP_E_SHAPE, "cbam: odd spatial kernel size up to 15 (got %d)", d.ks);
  EP_REQUIRE(cbam_side(d.N) > 0, EP_E_SHAPE, "cbam: N = %d is not a square token grid (the reference asserts the same)", d.N);
  EP_REQUIRE(!head || d.C > 0, EP_E_ARG, "cbam head: C must be positive");
  return 0;
}

static int cbam_params_ok(const ep_cbam_params* p, const char* what) {
  EP_REQUIRE(p, EP_E_ARG, "%s: null parameter struct", what);
  const float* ts[] = {p->fc1_w, p->fc2_w, p->conv_w, p->bn_w, p->bn_b};
  for (const float* t : ts) EP_REQUIRE(t && aligned16(t), EP_E_ALIGN, "%s: tensors must be non-null and 16-byte aligned", what);
  return 0;
}

static GemmParams cbg(const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f;
  g.extA = (int)lda; g.extB = (int)ldb;
  return g;
}

struct CbamTok { const void* x; int x_dtype; int64_t bstride; const int32_t* index; };
struct CbamBn { int training; float eps, momentum; float *running_mean, *running_var; int64_t* nbt; };

static int cbam_chan_table(const ep_cbam_dims& d, const CbamTok& t, float* tab, hipStream_t st) {
  const dim3 grid(d.B, (d.D / 4 + 255) / 256);
  if (t.x_dtype == EP_DTYPE_BF16)
    hipLaunchKernelGGL(ep_cbam_chan_kernel<true>, grid, dim3(256), 0, st, t.x, t.bstride, t.index, d.N, d.D, tab);
  else
    hipLaunchKernelGGL(ep_cbam_chan_kernel<false>, grid, dim3(256), 0, st, t.x, t.bstride, t.index, d.N, d.D, tab);
  EP_LAUNCH_CHECK("ep_cbam_chan_kernel");
  return 0;
}

template <int MODE>
static int cbam_rows(const ep_cbam_dims& d, const CbamTok& t, const float* vec, float* maps, int* arg, hipStream_t st) {
  const int cpl = (d.D / 4 + 63) / 64;
  const bool bf = t.x_dtype == EP_DTYPE_BF16;
#define EP_CB(C_)                                                                                                          \
  if (cpl <= C_) {                                                                                                         \
    if (bf) hipLaunchKernelGGL((ep_cbam_row_kernel<C_, true, MODE>), dim3(d.B), dim3(256), 0, st, t.x, t.bstride, t.index, \
                               d.N, d.D, vec, maps, arg);                                                                  \
    else hipLaunchKernelGGL((ep_cbam_row_kernel<C_, false, MODE>), dim3(d.B), dim3(256), 0, st, t.x, t.bstride, t.index,   \
                            d.N, d.D, vec, maps, arg);                                                                     \
    EP_LAUNCH_CHECK("ep_cbam_row_kernel");                                                                                 \
    return 0;                                                                                                              \
  }
  EP_CB(1) EP_CB(2) EP_CB(3) EP_CB(4) EP_CB(5) EP_CB(8) EP_CB(16)
#undef EP_CB
  set_error("cbam: D = %d too wide", d.D);
  return EP_E_UNSUPPORTED;
}

template <int MODE>
static int cbam_acc(const ep_cbam_dims& d, const CbamTok& t, const float* tw, const int* arg, const float* tab, const float* gc,
                    float* T, float* out, hipStream_t st) {
  const dim3 grid(d.B, (d.D / 4 + 255) / 256);
  if (t.x_dtype == EP_DTYPE_BF16)
    hipLaunchKernelGGL((ep_cbam_acc_kernel<true, MODE>), grid, dim3(256), 0, st, t.x, t.bstride, t.index, d.N, d.D, tw, arg, tab, gc, T,
                       out);
  else
    hipLaunchKernelGGL((ep_cbam_acc_kernel<false, MODE>), grid, dim3(256), 0, st, t.x, t.bstride, t.index, d.N, d.D, tw, arg, tab, gc, T,
                       out);
  EP_LAUNCH_CHECK("ep_cbam_acc_kernel");
  return 0;
}

// chan_table: optional cached (M, 3, D) table of ep_cbam_channel_table (rows addressed through the image index)
static int cbam_forward_core(const ep_cbam_dims& d, const CbamTok& t, const float* chan_table, const CbamBn& bn,
                             const ep_cbam_params& pr, const CbamWs& w, float* y, hipStream_t st) {
  const int D = d.D, B = d.B, N = d.N, rd = d.rd, side = cbam_side(N);
  const int64_t nd = (int64_t)B * D, nr = (int64_t)B * rd, nn = (int64_t)B * N;
  if (chan_table) {                                   // gather the batch's rows (B x 3 D, tiny)
    if (t.index) {
      hipLaunchKernelGGL(ep_cbam_gather_kernel, dim3((unsigned)((nd * 3 + 255) / 256)), dim3(256), 0, st, chan_table, t.index, nd * 3,
                         3 * D, w.tab);
    } else {
      EP_HIP(hipMemcpyAsync(w.tab, chan_table, (size_t)nd * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  } else {
    EP_TRY(cbam_chan_table(d, t, w.tab, st));
  }
  // channel gate: pre = fc2 (relu(fc1 avg) + relu(fc1 max))
  EP_TRY(gemm(true, true, cbg(w.tab, 3 * D, pr.fc1_w, D, w.ha, rd, B, rd, D), 1, st));
  EP_TRY(gemm(true, true, cbg(w.tab + D, 3 * D, pr.fc1_w, D, w.hm, rd, B, rd, D), 1, st));
  hipLaunchKernelGGL(ep_cbam_hsum_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, w.ha, w.hm, nr, w.hs);
  EP_TRY(gemm(true, true, cbg(w.hs, rd, pr.fc2_w, rd, w.pre, D, B, D, rd), 1, st));
  hipLaunchKernelGGL(ep_cbam_sigmoid_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, w.pre, nd, w.gc);
  EP_LAUNCH_CHECK("ep_cbam channel gate kernels");
  // spatial gate
  EP_TRY(cbam_rows<0>(d, t, w.gc, w.maps, w.arg, st));
  hipLaunchKernelGGL(ep_cbam_conv_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, w.maps, pr.conv_w, side, d.ks, nn, w.conv);
  const int bng = cbam_bn_groups(nn);
  if (bn.training) hipLaunchKernelGGL(ep_cbam_bn1_part_kernel, dim3(bng), dim3(1024), 0, st, w.conv, nn, w.red);
  hipLaunchKernelGGL(ep_cbam_bn1_kernel, dim3(bng), dim3(1024), 0, st, w.conv, nn, bn.training, bn.eps, bn.momentum, pr.bn_w, pr.bn_b,
                     bn.running_mean, bn.running_var, bn.nbt, w.red, w.st, w.gs);
  EP_LAUNCH_CHECK("ep_cbam spatial gate kernels");
  return cbam_acc<0>(d, t, w.gs, nullptr, w.tab, w.gc, w.T, y, st);
}

static int cbam_backward_core(const ep_cbam_dims& d, const CbamTok& t, const ep_cbam_params& pr, const float* dy,
                              const ep_cbam_params& gr, int acc, const CbamWs& w, hipStream_t st) {
  const int D = d.D, B = d.B, N = d.N, rd = d.rd, side = cbam_side(N);
  const int64_t nd = (int64_t)B * D, nr = (int64_t)B * rd, nn = (int64_t)B * N;
  const unsigned ed = (unsigned)((nd + 255) / 256);
  // out = R0 + gc T :  dT = dout gc
  hipLaunchKernelGGL(ep_cbam_dpre_kernel, dim3(ed), dim3(256), 0, st, dy, (const float*)nullptr, (const float*)nullptr, w.gc, nd,
                     (float*)nullptr, w.dT);
  EP_LAUNCH_CHECK("ep_cbam_dpre_kernel (dT)");
  EP_TRY(cbam_rows<1>(d, t, w.dT, w.dgs, nullptr, st));                                   // pass D: d gs
  const int bng = cbam_bn_groups(nn);
  hipLaunchKernelGGL(ep_cbam_bn1_bwd_part_kernel, dim3(bng), dim3(1024), 0, st, w.dgs, w.gs, w.conv, nn, w.st, w.red);
  hipLaunchKernelGGL(ep_cbam_bn1_bwd_kernel, dim3(bng), dim3(1024), 0, st, w.dgs, w.gs, w.conv, nn, w.st, pr.bn_w, acc, w.red, gr.bn_w,
                     gr.bn_b, w.dconv);
  const int nw = 2 * d.ks * d.ks;
  int wrs = B / 16; wrs = wrs < 1 ? 1 : (wrs > CBAM_WRS ? CBAM_WRS : wrs);
  hipLaunchKernelGGL(ep_cbam_conv_wgrad_kernel, dim3(nw, wrs), dim3(256), 0, st, w.dconv, w.maps, B, side, d.ks, w.wpart);
  hipLaunchKernelGGL(ep_cbam_conv_wfin_kernel, dim3((nw + 127) / 128), dim3(128), 0, st, w.wpart, nw, wrs, acc, gr.conv_w);
  hipLaunchKernelGGL(ep_cbam_conv_bwd_kernel, dim3((unsigned)((2 * nn + 255) / 256)), dim3(256), 0, st, w.dconv, pr.conv_w, side, d.ks,
                     2 * nn, w.dmaps);
  EP_LAUNCH_CHECK("ep_cbam spatial backward kernels");
  EP_TRY(cbam_acc<1>(d, t, w.dmaps, w.arg, nullptr, nullptr, nullptr, w.E, st));          // pass E
  // d gc = dout T + E ;  dpre = d gc gc (1 - gc) ;  pre = (relu(ha) + relu(hm)) fc2^T
  hipLaunchKernelGGL(ep_cbam_dpre_kernel, dim3(ed), dim3(256), 0, st, dy, w.T, w.E, w.gc, nd, w.dpre, (float*)nullptr);
  EP_LAUNCH_CHECK("ep_cbam_dpre_kernel");
  { GemmParams g = cbg(w.dpre, D, w.hs, rd, gr.fc2_w, rd, D, rd, B); g.accumulate = acc; EP_TRY(gemm(false, false, g, 1, st)); }   // d fc2 = dpre^T hs
  { GemmParams g = cbg(w.dpre, D, pr.fc2_w, rd, w.dh, rd, B, rd, D); g.extB = rd; EP_TRY(gemm(true, false, g, 1, st)); }            // dh = dpre fc2
  hipLaunchKernelGGL(ep_cbam_dh_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, w.dh, w.ha, w.hm, nr, w.dha, w.dhm);
  EP_LAUNCH_CHECK("ep_cbam_dh_kernel");
  { GemmParams g = cbg(w.dha, rd, w.tab, 3 * D, gr.fc1_w, D, rd, D, B); g.extB = D; g.accumulate = acc; EP_TRY(gemm(false, false, g, 1, st)); }
  { GemmParams g = cbg(w.dhm, rd, w.tab + D, 3 * D, gr.fc1_w, D, rd, D, B); g.extB = D; g.accumulate = 1; EP_TRY(gemm(false, false, g, 1, st)); }
  return 0;
}

static ep_cbam_params cbam_views(float* base, const int64_t o[CBAM_NT]) {
  ep_cbam_params p;
  p.fc1_w = base + o[0]; p.fc2_w = base + o[1]; p.conv_w = base + o[2]; p.bn_w = base + o[3]; p.bn_b = base + o[4];
  return p;
}

}  // namespace ep

using namespace ep;

extern "C" {

int ep_cbam_channel_table(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                          float* table, ep_stream_t stream) {
  EP_TRY(check_tokens(x, x_dtype, x_bstride, B, N, D, 1));
  EP_REQUIRE(table && aligned16(table), EP_E_ARG, "ep_cbam_channel_table: output null or not 16-byte aligned");
  ep_cbam_dims d{}; d.B = B; d.N = N; d.D = D;
  return cbam_chan_table(d, CbamTok{x, x_dtype, x_bstride, image_index}, table, (hipStream_t)stream);
}

size_t ep_cbam_pool_workspace_bytes(const ep_cbam_dims* dims) {
  if (!dims || cbam_check(*dims, false) != 0) return 0;
  return cbam_carve(*dims, nullptr, false).total;
}

int ep_cbam_pool_forward(const ep_cbam_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                         const float* channel_table, int training, float bn_eps, float bn_momentum, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, const ep_cbam_params* params, float* y, void* ws,
                         size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && y && ws, EP_E_ARG, "ep_cbam_pool_forward: null pointer");
  EP_TRY(cbam_check(*dims, false));
  EP_TRY(cbam_params_ok(params, "ep_cbam_pool_forward"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, 1));
  EP_REQUIRE(aligned16(ws) && aligned16(y), EP_E_ALIGN, "ep_cbam_pool_forward: y / ws must be 16-byte aligned");
  EP_REQUIRE(training || (running_mean && running_var), EP_E_ARG, "ep_cbam_pool_forward: eval needs the running statistics");
  const CbamWs w = cbam_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cbam_pool_forward: workspace %zu < %zu", ws_bytes, w.total);
  const CbamBn bn{training, bn_eps, bn_momentum, running_mean, running_var, num_batches_tracked};
  return cbam_forward_core(*dims, CbamTok{x, x_dtype, x_bstride, image_index}, channel_table, bn, *params, w, y, (hipStream_t)stream);
}

int ep_cbam_pool_backward(const ep_cbam_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                          const ep_cbam_params* params, const float* dy, const ep_cbam_params* grads, int accumulate, void* ws,
                          size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(dims && dy && ws, EP_E_ARG, "ep_cbam_pool_backward: null pointer");
  EP_TRY(cbam_check(*dims, false));
  EP_TRY(cbam_params_ok(params, "ep_cbam_pool_backward(params)"));
  EP_TRY(cbam_params_ok(grads, "ep_cbam_pool_backward(grads)"));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, dims->B, dims->N, dims->D, 1));
  EP_REQUIRE(aligned16(ws) && aligned16(dy), EP_E_ALIGN, "ep_cbam_pool_backward: dy / ws must be 16-byte aligned");
  const CbamWs w = cbam_carve(*dims, ws, false);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cbam_pool_backward: workspace %zu < %zu", ws_bytes, w.total);
  return cbam_backward_core(*dims, CbamTok{x, x_dtype, x_bstride, image_index}, *params, dy, *grads, accumulate, w,
                            (hipStream_t)stream);
}

int64_t ep_cbam_head_param_offsets(const ep_cbam_dims* dims, int64_t offsets[7]) { return cbam_offsets(*dims, offsets); }

size_t ep_cbam_head_workspace_bytes(const ep_cbam_dims* dims) {
  if (!dims || cbam_check(*dims, true) != 0) return 0;
  return cbam_carve(*dims, nullptr, true).total;
}

int ep_cbam_head_train_step(const ep_cbam_step* s, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(s && ws, EP_E_ARG, "ep_cbam_head_train_step: null pointer");
  const ep_cbam_dims& d = s->dims;
  EP_TRY(cbam_check(d, true));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "workspace must be 16-byte aligned");
  const CbamWs w = cbam_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cbam_head_train_step: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(s->params && s->grads, EP_E_ARG, "params / grads null");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CBAM_NT];
  const int64_t total = cbam_offsets(d, offs);
  const ep_cbam_params pr = cbam_views(s->params, offs), gr = cbam_views(s->grads, offs);
  float* Wc = s->params + offs[5]; float* bc = s->params + offs[6];
  if (s->phases & 1) {
    EP_REQUIRE(s->x && s->targets && s->running_mean && s->running_var && s->stats && s->tok_running_mean && s->tok_running_var,
               EP_E_ARG, "train step: null input");
    EP_TRY(check_tokens(s->x, s->x_dtype, s->x_bstride, d.B, d.N, d.D, 1));
    const CbamTok t{s->x, s->x_dtype, s->x_bstride, s->image_index};
    const CbamBn bn{1, s->tok_bn_eps, s->tok_bn_momentum, s->tok_running_mean, s->tok_running_var, s->tok_num_batches_tracked};
    EP_TRY(cbam_forward_core(d, t, s->image_stats, bn, pr, w, w.y, st));
    EP_TRY(bn_forward_train(w.y, d.B, d.D, s->bn_eps, s->bn_momentum, w.z, w.rstd, s->running_mean, s->running_var,
                            s->num_batches_tracked, w.bnpart, st));
    EP_TRY(linear_forward(w.z, Wc, bc, d.B, d.D, d.C, w.logits, w.ldl, st));
    EP_TRY(cross_entropy(w.logits, w.ldl, s->targets, d.B, d.C, s->grad_scale, nullptr, w.dlogits, w.rowstat, st));
    EP_TRY(ce_stats(w.rowstat, d.B, s->stats, st));
    EP_TRY(linear_backward(w.dlogits, w.ldl, w.z, Wc, d.B, d.D, d.C, w.dz, s->grads + offs[5], s->grads + offs[6],
                           s->accumulate, st));
    EP_TRY(bn_backward(w.dz, w.z, w.rstd, d.B, d.D, w.dy, w.bnpart, st));
    EP_TRY(cbam_backward_core(d, t, pr, w.dy, gr, s->accumulate, w, st));
  }
  if (s->phases & 2) {
    EP_REQUIRE(s->found_inf, EP_E_ARG, "optimizer phase needs found_inf");
    int64_t sizes[CBAM_NT];
    cbam_sizes(d, sizes);
    const int trust[CBAM_NT] = {1, 1, 1, 0, 0, 1, 0};            // util/lars.py:22: ndim > 1 (the conv weights are 4-D)
    ep_segment segs[CBAM_NT];
    for (int i = 0; i < CBAM_NT; ++i) segs[i] = ep_segment{offs[i], sizes[i], trust[i], 0};
    EP_TRY(optim_step(s->optimizer, s->params, s->grads, s->opt_state0, s->opt_state1, total,
                      s->optimizer == 0 ? segs : nullptr, s->optimizer == 0 ? CBAM_NT : 0, s->lr, s->weight_decay, s->momentum,
                      s->trust_coefficient, s->inv_scale, s->beta1, s->beta2, s->adam_eps, s->opt_step, s->found_inf,
                      s->grad_norm, w.opt_ws, w.opt_ws_bytes, st));
  }
  return 0;
}

int ep_cbam_head_eval_forward(const ep_cbam_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                              const float* channel_table, float tok_bn_eps, const float* tok_running_mean,
                              const float* tok_running_var, const float* params, const float* running_mean,
                              const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                              ep_stream_t stream) {
  EP_REQUIRE(dims && x && params && running_mean && running_var && tok_running_mean && tok_running_var && logits && ws, EP_E_ARG,
             "ep_cbam_head_eval_forward: null pointer");
  const ep_cbam_dims& d = *dims;
  EP_TRY(cbam_check(d, true));
  EP_TRY(check_tokens(x, x_dtype, x_bstride, d.B, d.N, d.D, 1));
  const CbamWs w = cbam_carve(d, ws, true);
  EP_REQUIRE(ws_bytes >= w.total, EP_E_WORKSPACE, "ep_cbam_head_eval_forward: workspace %zu < %zu", ws_bytes, w.total);
  EP_REQUIRE(ldl >= d.C, EP_E_ARG, "ldl < C");
  hipStream_t st = (hipStream_t)stream;
  int64_t offs[CBAM_NT];
  cbam_offsets(d, offs);
  const ep_cbam_params pr = cbam_views(const_cast<float*>(params), offs);
  const CbamBn bn{0, tok_bn_eps, 0.f, const_cast<float*>(tok_running_mean), const_cast<float*>(tok_running_var), nullptr};
  EP_TRY(cbam_forward_core(d, CbamTok{x, x_dtype, x_bstride, image_index}, channel_table, bn, pr, w, w.y, st));
  EP_TRY(bn_forward_eval(w.y, d.B, d.D, bn_eps, running_mean, running_var, w.z, st));
  return linear_forward(w.z, params + offs[5], params + offs[6], d.B, d.D, d.C, logits, ldl, st);
}

}  // extern "C"
