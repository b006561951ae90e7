// Weighted k-NN classifier on frozen features (reference engine_finetune.py:224-266 knn_classifier, driven by
// main_linprobe.py:411-465): similarity = test . train^T (exact-fp32 MFMA contraction, ep_gemm.hip), the k largest
// similarities per test row (exact radix select, this file), votes exp(sim / T) per neighbour label and the five
// best classes per row.  The neighbour lists are computed once for the largest k and every smaller k / every
// temperature of the reference's sweep reads a prefix of them.
#include <math.h>
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

// order-preserving map float -> uint32 (ascending)
__device__ __forceinline__ uint32_t f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// x <- x / max(||x||_2, eps) per row (torch.nn.functional.normalize, main_linprobe.py:441-442); one wave per row
__global__ __launch_bounds__(256) void ep_l2_normalize_kernel(const float* __restrict__ x, int64_t rows, int D, float eps,
                                                            float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[r * D + d]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), eps);
  for (int d = lane; d < D; d += 64) out[r * D + d] = x[r * D + d] * inv;
}

// ---------------------------------------------------------------------------------------------
// exact top-k of one row per workgroup: three radix passes (11 + 11 + 10 bits) find the key of the k-th
// largest element, a fourth pass gathers everything above it plus the lowest-index elements equal to it,
// a bitonic sort orders the k survivors by (value descending, index ascending).
// ---------------------------------------------------------------------------------------------
constexpr int KNN_T = 256;          // threads per row
constexpr int KNN_CAP = 1024;       // max k
constexpr int KNN_EQCAP = 2048;     // elements equal to the threshold that are ranked by index in LDS

struct SelState { int bin; int need; };

// From hist[0..nb) (counts per bin, bin = larger key means larger value): the highest bin b such that
// count(bins > b) < need <= count(bins >= b); returns b and need - count(bins > b).
__device__ __forceinline__ SelState pick_bin(const int* hist, int nb, int need, int* scratch) {
  const int tid = threadIdx.x;
  const int per = nb / KNN_T;                       // bins per thread (8 or 4)
  int local = 0;
  const int hi = nb - 1 - tid * per;                // this thread owns bins hi, hi-1, ..., hi-per+1
  for (int j = 0; j < per; ++j) local += hist[hi - j];
  scratch[tid] = local;
  __syncthreads();
  if (tid == 0) {                                   // 256-long serial scan: ~1 us, once per pass
    int run = 0;
    for (int t = 0; t < KNN_T; ++t) { const int c = scratch[t]; scratch[t] = run; run += c; }
  }
  __syncthreads();
  const int above = scratch[tid];
  __shared__ SelState res;
  if (above < need && need <= above + local) {
    int run = above;
    for (int j = 0; j < per; ++j) {
      const int c = hist[hi - j];
      if (need <= run + c) { res.bin = hi - j; res.need = need - run; break; }
      run += c;
    }
  }
  __syncthreads();
  const SelState r = res;
  __syncthreads();
  return r;
}

// Fast path (round 4): the three histogram passes above cost one LDS atomic per element and pass -- with cosine similarities
// nearly every element of a row falls into the same few top-bit bins, the atomics serialise, and a 1.28 M-element row took
// ~7 ms (half of the whole search).  Instead: pass A streams the row once with vector loads and keeps, per thread, the
// maxima of 16 interleaved element groups (4096 group maxima per row, no LDS traffic); their r-th largest is a threshold t
// above which an expected C ~ max(3 k, 800) elements lie (P(group maximum > t) = 1 - exp(-C / 4096) for randomly placed
// elements); pass B streams the row again and gathers the elements above t (one LDS atomic per CANDIDATE).  If between k and
// 4096 came together, the k largest of them by (value descending, index ascending) ARE the row's top k -- exact, same tie
// rule; otherwise (adversarial orderings, massive ties) the radix passes below run as before.
constexpr int KNN_G = 16;           // element groups per thread
constexpr int KNN_FCAP = 4096;      // candidate capacity of the fast path (= KNN_T * KNN_G group maxima)

__global__ __launch_bounds__(KNN_T) void ep_knn_select_kernel(const float* __restrict__ Cm, int64_t ldc, int n, int k,
                                                            float* __restrict__ sims, int32_t* __restrict__ idx_out,
                                                            int out_ld, int fast) {
  __shared__ uint32_t buf[2 * KNN_FCAP];            // 32 KiB, shared by both paths
  __shared__ int fcount;
  int* hist = reinterpret_cast<int*>(buf);          // [2048]                radix path
  uint32_t* ckey = buf + 2048;                      // [KNN_CAP]
  int* cidx = reinterpret_cast<int*>(buf + 3072);   // [KNN_CAP]
  int* eqidx = reinterpret_cast<int*>(buf + 4096);  // [KNN_EQCAP]
  int* scratch = reinterpret_cast<int*>(buf + 6144);// [KNN_T]
  int* counters = reinterpret_cast<int*>(buf + 6400);   // [0] = #gathered above the threshold, [1] = #equal seen
  const int tid = threadIdx.x;
  const float* row = Cm + (int64_t)blockIdx.x * ldc;

  if (fast && n >= 8 * KNN_FCAP) {
    uint32_t* fkey = buf;                            // [KNN_FCAP] group maxima, then candidate keys
    int* fidx = reinterpret_cast<int*>(buf + KNN_FCAP);
    const f4* row4 = reinterpret_cast<const f4*>(row);
    const int n4 = n >> 2;
    uint32_t gm[KNN_G];
#pragma unroll
    for (int g = 0; g < KNN_G; ++g) gm[g] = 0u;
    for (int base = 0; base < n4; base += KNN_T * KNN_G) {
#pragma unroll
      for (int g = 0; g < KNN_G; ++g) {
        const int i4 = base + g * KNN_T + tid;
        if (i4 < n4) {
          const f4 v = row4[i4];
          const uint32_t a = f2key(v.x), b2 = f2key(v.y), c = f2key(v.z), d = f2key(v.w);
          const uint32_t m01 = a > b2 ? a : b2, m23 = c > d ? c : d;
          const uint32_t mx = m01 > m23 ? m01 : m23;
          gm[g] = gm[g] > mx ? gm[g] : mx;
        }
      }
    }
    if (tid == 0)
      for (int i = n4 * 4; i < n; ++i) { const uint32_t key = f2key(row[i]); gm[0] = gm[0] > key ? gm[0] : key; }
#pragma unroll
    for (int g = 0; g < KNN_G; ++g) fkey[g * KNN_T + tid] = gm[g];
    __syncthreads();
    // descending bitonic sort of the 4096 group maxima
    for (int sz = 2; sz <= KNN_FCAP; sz <<= 1)
      for (int st = sz >> 1; st > 0; st >>= 1) {
        for (int i = tid; i < KNN_FCAP; i += KNN_T) {
          const int j = i ^ st;
          if (j > i) {
            const bool up = (i & sz) == 0;
            const uint32_t ka = fkey[i], kb = fkey[j];
            if ((ka < kb) == up) { fkey[i] = kb; fkey[j] = ka; }
          }
        }
        __syncthreads();
      }
    float want = 3.0f * (float)k;
    want = want < 800.f ? 800.f : (want > 3072.f ? 3072.f : want);
    int r = (int)((float)KNN_FCAP * (1.0f - __expf(-want / (float)KNN_FCAP)));
    r = r < 1 ? 1 : (r > KNN_FCAP - 1 ? KNN_FCAP - 1 : r);
    const uint32_t t = fkey[r - 1];                 // the r-th largest group maximum
    if (tid == 0) fcount = 0;
    __syncthreads();                                 // (everyone has read t; the key array is free)
    for (int i4 = tid; i4 < n4; i4 += KNN_T) {
      const f4 v = row4[i4];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t key = f2key(e[j]);
        if (key > t) {
          const int pos = atomicAdd(&fcount, 1);
          if (pos < KNN_FCAP) { fkey[pos] = key; fidx[pos] = 4 * i4 + j; }
        }
      }
    }
    if (tid == 0)
      for (int i = n4 * 4; i < n; ++i) {
        const uint32_t key = f2key(row[i]);
        if (key > t) { const int pos = atomicAdd(&fcount, 1); if (pos < KNN_FCAP) { fkey[pos] = key; fidx[pos] = i; } }
      }
    __syncthreads();
    const int cnt = fcount;
    if (cnt >= k && cnt <= KNN_FCAP) {
      int m = 1; while (m < cnt) m <<= 1;
      for (int i = cnt + tid; i < m; i += KNN_T) { fkey[i] = 0u; fidx[i] = 0x7fffffff; }
      __syncthreads();
      for (int sz = 2; sz <= m; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
          for (int i = tid; i < m; i += KNN_T) {
            const int j = i ^ st;
            if (j > i) {
              const bool up = (i & sz) == 0;         // "up" segments end up descending by value, ascending by index
              const uint32_t ka = fkey[i], kb = fkey[j];
              const int ia = fidx[i], ib = fidx[j];
              const bool a_after_b = (ka < kb) || (ka == kb && ia > ib);
              if (a_after_b == up) { fkey[i] = kb; fkey[j] = ka; fidx[i] = ib; fidx[j] = ia; }
            }
          }
          __syncthreads();
        }
      for (int i = tid; i < k; i += KNN_T) {
        sims[(int64_t)blockIdx.x * out_ld + i] = key2f(fkey[i]);
        idx_out[(int64_t)blockIdx.x * out_ld + i] = fidx[i];
      }
      return;
    }
    __syncthreads();                                 // fall through: the radix passes (they re-read the row)
  }

  auto clear_hist = [&](int nb) { for (int i = tid; i < nb; i += KNN_T) hist[i] = 0; __syncthreads(); };
  // pass 1: top 11 bits
  clear_hist(2048);
  for (int i = tid; i < n; i += KNN_T) atomicAdd(&hist[f2key(row[i]) >> 21], 1);
  __syncthreads();
  const SelState s1 = pick_bin(hist, 2048, k, scratch);
  // pass 2: next 11 bits among keys with the chosen top bits
  clear_hist(2048);
  for (int i = tid; i < n; i += KNN_T) {
    const uint32_t key = f2key(row[i]);
    if ((int)(key >> 21) == s1.bin) atomicAdd(&hist[(key >> 10) & 2047], 1);
  }
  __syncthreads();
  const SelState s2 = pick_bin(hist, 2048, s1.need, scratch);
  const uint32_t pre22 = ((uint32_t)s1.bin << 11) | (uint32_t)s2.bin;
  // pass 3: last 10 bits
  clear_hist(1024);
  for (int i = tid; i < n; i += KNN_T) {
    const uint32_t key = f2key(row[i]);
    if ((key >> 10) == pre22) atomicAdd(&hist[key & 1023], 1);
  }
  __syncthreads();
  const SelState s3 = pick_bin(hist, 1024, s2.need, scratch);
  const uint32_t thr = (pre22 << 10) | (uint32_t)s3.bin;      // key of the k-th largest element
  const int need_eq = s3.need;                                  // how many elements equal to it belong to the top k
  const int n_gt = k - need_eq;
  // pass 4: gather
  if (tid < 2) counters[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += KNN_T) {
    const uint32_t key = f2key(row[i]);
    if (key > thr) {
      const int p = atomicAdd(&counters[0], 1);
      ckey[p] = key; cidx[p] = i;
    } else if (key == thr) {
      const int p = atomicAdd(&counters[1], 1);
      if (p < KNN_EQCAP) eqidx[p] = i;
    }
  }
  __syncthreads();
  const int n_eq = counters[1];
  if (n_eq <= KNN_EQCAP) {
    // rank the equal elements by index (bitonic sort of up to 2048 ints), keep the lowest need_eq
    int m = 1; while (m < n_eq) m <<= 1;
    for (int i = n_eq + tid; i < m; i += KNN_T) eqidx[i] = 0x7fffffff;
    __syncthreads();
    for (int sz = 2; sz <= m; sz <<= 1)
      for (int st = sz >> 1; st > 0; st >>= 1) {
        for (int i = tid; i < m; i += KNN_T) {
          const int j = i ^ st;
          if (j > i) {
            const bool up = (i & sz) == 0;
            const int a = eqidx[i], b = eqidx[j];
            if ((a > b) == up) { eqidx[i] = b; eqidx[j] = a; }
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < need_eq; i += KNN_T) { ckey[n_gt + i] = thr; cidx[n_gt + i] = eqidx[i]; }
  } else if (tid < 64) {
    // pathological row (thousands of exact ties at the threshold): one wave walks the row in index order
    int taken = 0;
    for (int base = 0; base < n && taken < need_eq; base += 64) {
      const int i = base + tid;
      const bool hit = i < n && f2key(row[i]) == thr;
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
      const int before = __builtin_popcountll(mask & ((1ull << tid) - 1ull));
      if (hit && taken + before < need_eq) { ckey[n_gt + taken + before] = thr; cidx[n_gt + taken + before] = i; }
      taken += __builtin_popcountll(mask);
    }
  }
  __syncthreads();
  // order the k survivors: value descending, index ascending
  int m = 1; while (m < k) m <<= 1;
  for (int i = k + tid; i < m; i += KNN_T) { ckey[i] = 0u; cidx[i] = 0x7fffffff; }
  __syncthreads();
  for (int sz = 2; sz <= m; sz <<= 1)
    for (int st = sz >> 1; st > 0; st >>= 1) {
      for (int i = tid; i < m; i += KNN_T) {
        const int j = i ^ st;
        if (j > i) {
          const bool up = (i & sz) == 0;             // "up" segments end up descending by value
          const uint32_t ka = ckey[i], kb = ckey[j];
          const int ia = cidx[i], ib = cidx[j];
          const bool a_after_b = (ka < kb) || (ka == kb && ia > ib);
          if (a_after_b == up) { ckey[i] = kb; ckey[j] = ka; cidx[i] = ib; cidx[j] = ia; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < k; i += KNN_T) {
    sims[(int64_t)blockIdx.x * out_ld + i] = key2f(ckey[i]);
    idx_out[(int64_t)blockIdx.x * out_ld + i] = cidx[i];
  }
}

// votes of the k nearest neighbours (engine_finetune.py:246-256): probs[c] = sum_{i<k, label_i = c} exp(sim_i / T);
// the five best classes (ties: lowest class first, like a stable descending sort) and the hit counters
// (engine_finetune.py:259-262).  One wave per test row; lane l owns the classes c = l mod 64.
__global__ __launch_bounds__(64) void ep_knn_vote_kernel(const float* __restrict__ sims, const int32_t* __restrict__ idx,
                                                       int ld, const int64_t* __restrict__ train_labels, int k, float T,
                                                       int num_classes, const int64_t* __restrict__ targets,
                                                       int32_t* __restrict__ pred, float* __restrict__ counts) {
  extern __shared__ float probs[];
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int c = lane; c < num_classes; c += 64) probs[c] = 0.f;
  __syncthreads();
  for (int i = 0; i < k; ++i) {
    const int cls = (int)train_labels[idx[(int64_t)r * ld + i]];
    const float w = expf(sims[(int64_t)r * ld + i] / T);
    if (cls >= 0 && cls < num_classes && (cls & 63) == lane) probs[cls] += w;
  }
  __syncthreads();
  const int tgt = targets ? (int)targets[r] : -1;
  int hit1 = 0, hit5 = 0;
  for (int j = 0; j < 5 && j < num_classes; ++j) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < num_classes; c += 64) { const float v = probs[c]; if (v > best) { best = v; bi = c; } }
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { pred[(int64_t)r * 5 + j] = bi; probs[bi] = -INFINITY; }
    if (bi == tgt) { hit5 = 1; if (j == 0) hit1 = 1; }
    __syncthreads();
  }
  if (lane == 0 && targets && counts) {
    if (hit1) atomicAdd(&counts[0], 1.0f);           // integer-valued float counters: exact and order independent
    if (hit5) atomicAdd(&counts[1], 1.0f);
  }
}

static int64_t knn_ld(int n_train) { return ((int64_t)n_train + 3) / 4 * 4; }
static int knn_chunk_rows(int M, int n_train) {
  static int64_t cap = -1;                                            // EP_KNN_CHUNK_GIB (default 4): similarities held at a time
  if (cap < 0) { const char* e = getenv("EP_KNN_CHUNK_GIB"); cap = (int64_t)(e ? atoi(e) : 4) << 30; if (cap < (1LL << 28)) cap = 1LL << 28; }
  int64_t rows = cap / (knn_ld(n_train) * 4);
  if (rows >= 128) rows -= rows % 64;                                 // whole 64-row tiles of the similarity contraction
  if (rows < 16) rows = 16;
  if (rows > M) rows = M;
  return (int)rows;
}

}  // namespace ep

using namespace ep;

extern "C" {

int ep_l2_normalize(const float* x, int64_t rows, int D, float eps, float* out, ep_stream_t stream) {
  EP_REQUIRE(x && out && rows > 0 && D > 0, EP_E_ARG, "ep_l2_normalize: bad argument");
  hipLaunchKernelGGL(ep_l2_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, D,
                     eps, out);
  EP_LAUNCH_CHECK("ep_l2_normalize_kernel");
  return 0;
}

// The gallery is the same for every query chunk: its bf16 planes (ep_planes.hip) are split ONCE per search, and the
// similarity contractions run on the planes kernel (fp32 accuracy, 1.9 x the rate of splitting both operands on the fly).
// The workspace query cannot see D, so the planes region is sized for rows of up to KNN_PLANES_MAXD values; wider features
// keep the on-the-fly tile.  EP_KNN_PLANES=0 switches it off.
constexpr int KNN_PLANES_MAXD = 1536;
static bool knn_planes_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EP_KNN_PLANES"); on = e ? atoi(e) : 1; }
  return on != 0;
}
static size_t knn_sims_bytes(int M, int n_train) {
  return round_up((size_t)knn_chunk_rows(M, n_train) * knn_ld(n_train) * sizeof(float), 256);
}
static bool knn_planes_shape_ok(int n_train, int D) {
  return knn_planes_on() && n_train >= 4096 && n_train <= 64 * 65535 && D >= 32 && D <= KNN_PLANES_MAXD && D % 4 == 0;
}
size_t ep_knn_workspace_bytes_ex(int M, int n_train, int D) {
  if (M <= 0 || n_train <= 0 || D <= 0) return 0;
  size_t b = knn_sims_bytes(M, n_train);
  if (knn_planes_shape_ok(n_train, D)) b += round_up(planes_elems(n_train, D) * sizeof(uint16_t), 256);
  return b;
}
size_t ep_knn_workspace_bytes(int M, int n_train) { return ep_knn_workspace_bytes_ex(M, n_train, KNN_PLANES_MAXD); }

int ep_knn_topk(const float* test, const float* train, int M, int n_train, int D, int k, float* sims, int32_t* idx,
                int out_ld, void* ws, size_t ws_bytes, ep_stream_t stream) {
  EP_REQUIRE(test && train && sims && idx && ws, EP_E_ARG, "ep_knn_topk: null pointer");
  EP_REQUIRE(M > 0 && n_train > 0 && D > 0, EP_E_ARG, "ep_knn_topk: sizes must be positive");
  EP_REQUIRE(k >= 1 && k <= KNN_CAP && k <= n_train && out_ld >= k, EP_E_ARG, "ep_knn_topk: need 1 <= k <= min(%d, n_train) and out_ld >= k (k=%d)", KNN_CAP, k);
  EP_REQUIRE(ws_bytes >= knn_sims_bytes(M, n_train), EP_E_WORKSPACE, "ep_knn_topk: workspace %zu < %zu", ws_bytes,
             knn_sims_bytes(M, n_train));
  EP_REQUIRE(aligned16(ws), EP_E_ALIGN, "ep_knn_topk: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t ld = knn_ld(n_train);
  const int rows = knn_chunk_rows(M, n_train);
  float* Cm = static_cast<float*>(ws);
  static int fast_select = -1;                       // EP_KNN_FAST=0: the radix passes only
  if (fast_select < 0) { const char* e = getenv("EP_KNN_FAST"); fast_select = e ? atoi(e) : 1; }
  uint16_t* planes = nullptr;
  // (a caller that sized the workspace without the planes region -- an older ep_knn_workspace_bytes -- keeps the on-the-fly tile)
  if (knn_planes_shape_ok(n_train, D) && M >= 64 && aligned16(test) && aligned16(train) &&
      ws_bytes >= ep_knn_workspace_bytes_ex(M, n_train, D)) {
    planes = reinterpret_cast<uint16_t*>(static_cast<char*>(ws) + knn_sims_bytes(M, n_train));
    const PlaneSpec sp{train, n_train, D, D, planes, nullptr};
    EP_TRY(planes_split(&sp, 1, st));
  }
  for (int m0 = 0; m0 < M; m0 += rows) {
    const int mc = (M - m0) < rows ? (M - m0) : rows;
    GemmParams g{};
    g.A = test + (int64_t)m0 * D; g.lda = D; g.B = train; g.ldb = D; g.C = Cm; g.ldc = ld;
    g.M = mc; g.N = n_train; g.K = D; g.alpha = 1.f; g.extA = D; g.extB = D;
    if (planes) {
      g.Bpl = planes; g.ldbp = (int64_t)round_up((size_t)D, 32); g.pl_term = (int64_t)n_train * g.ldbp;
      g.m_fast = 1;                                  // the few query tiles of a gallery tile run together: its planes come from HBM once
      EP_TRY(gemm_planes(g, 1, st));
    } else
    EP_TRY(gemm(true, true, g, 1, st));                         // similarity = features @ train_features.t()  (:243)
    hipLaunchKernelGGL(ep_knn_select_kernel, dim3(mc), dim3(KNN_T), 0, st, Cm, ld, n_train, k, sims + (int64_t)m0 * out_ld,
                       idx + (int64_t)m0 * out_ld, out_ld, fast_select);   // similarity.topk(k, largest=True, sorted=True)   (:244)
    EP_LAUNCH_CHECK("ep_knn_select_kernel");
  }
  return 0;
}

int ep_knn_vote(const float* sims, const int32_t* idx, int ld, const int64_t* train_labels, int M, int k, float T,
                int num_classes, const int64_t* targets, int32_t* pred, float* counts, ep_stream_t stream) {
  EP_REQUIRE(sims && idx && train_labels && pred, EP_E_ARG, "ep_knn_vote: null pointer");
  EP_REQUIRE(M > 0 && k >= 1 && k <= ld && num_classes >= 1 && num_classes <= 15000 && T > 0.f, EP_E_ARG, "ep_knn_vote: bad sizes");
  hipLaunchKernelGGL(ep_knn_vote_kernel, dim3(M), dim3(64), (size_t)num_classes * 4, (hipStream_t)stream, sims, idx, ld,
                     train_labels, k, T, num_classes, targets, pred, counts);
  EP_LAUNCH_CHECK("ep_knn_vote_kernel");
  return 0;
}

}  // extern "C"
