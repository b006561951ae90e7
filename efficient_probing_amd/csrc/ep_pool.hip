// EP attentive pooling: the two streaming passes over the frozen tokens (gfx950 / CDNA4).
//
//   forward  (reference poolings/ep.py:35-44):  S = (cls*scale) x^T ; A = softmax_n S ; P = A x
//   backward (autograd of the same lines)     :  dA = dP x^T ; dS = A (dA - delta) ;
//                                                dcls = scale * sum_b dS x
//
// Design (DESIGN.md section "pool kernels"):
//   * one workgroup streams whole images; the image's tokens are copied HBM -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of NSLOT tiles of TT
//     tokens, several tiles ahead of the compute, so x is read from HBM exactly once per pass
//     and never touches a VGPR on the way in;
//   * the Q queries are split over the NW waves of the workgroup (QW queries per wave); every
//     wave reads every token row from LDS (lane = 16-byte chunk of the row, conflict-free
//     ds_read_b128) and owns the FULL D-dimension for its queries, so there is no cross-wave
//     reduction at all -- the only synchronisation is one s_barrier per tile for the ring;
//   * scores: lane-local FMAs, then a v_permlane32_swap / v_permlane16_swap / DPP butterfly that
//     leaves score (q, t) replicated in the 16 lanes of row t of register q;
//   * softmax: lazy-max online softmax evaluated lane-parallel (row = token), weights broadcast
//     to SGPRs with v_readlane and used as the scalar operand of the pooling FMAs.
//   * LDS-DMA completion is tracked with counted s_waitcnt vmcnt(N) (never 0 in steady state).
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int TB = 4;                 // tokens per butterfly mini-batch (one per 16-lane row)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LAZY_MAX_THR = 12.0f; // rescale only when a score exceeds the running max by this

// s_waitcnt vmcnt(n) with a runtime (wave-uniform) n in [0, 63]
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

__device__ __forceinline__ void ring_barrier() {
  // all of this wave's LDS reads of the previous tile have retired (their results were
  // consumed), the wave's own DMA pieces of the next tile have landed (wait_vmcnt before).
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Issue the LDS-DMA copy of one ring item: `rows` rows of D floats starting at `src`
// (rows*D*4 valid bytes, the rest of the slot is filled with duplicates of the last chunk).
template <int NW>
__device__ __forceinline__ void dma_item(const char* src, int valid_bytes, char* slot, int npiece,
                                         int kdma, int w, int lane) {
  const int limit = valid_bytes - 16;
  for (int j = 0; j < kdma; ++j) {
    int pc = w + NW * j;                       // wave-uniform piece index
    pc = pc < npiece ? pc : npiece - 1;        // surplus instructions re-copy the last piece
    int off = pc * 1024 + lane * 16;
    off = off < limit ? off : limit;
    char* dst = slot + pc * 1024;              // wave-uniform LDS base; HW adds lane*16
    __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lds_ptr_t)dst, 16, 0, 0);
  }
}

// butterfly reduction of V = 4*QW lane-partial values; on return u[q] holds, in every lane of
// row t (lanes 16t..16t+15), the full 64-lane sum of part[q][t].
template <int QW>
__device__ __forceinline__ void butterfly(const float (&part)[QW][TB], float (&u)[QW]) {
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    // fold32(a,b): lanes<32 <- a, lanes>=32 <- b.  fold16(r0,r1): rows <- [r0.lo, r1.lo, r0.hi, r1.hi]
    // want rows [t0,t1,t2,t3]  =>  r0 = fold32(t0,t2), r1 = fold32(t1,t3)
    float r0 = fold32(part[q][0], part[q][2]);
    float r1 = fold32(part[q][1], part[q][3]);
    u[q] = row16_sum(fold16(r0, r1));
  }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int QW, int KP, int NW, int TT>
__global__ __launch_bounds__(NW * 64, 2) void ep_pool_fwd_kernel(PoolParams p) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int rowbytes = D * 4;
  const int nchunk = D >> 2;                    // 16-byte chunks per row
  const int npiece = p.slot_bytes >> 10;
  const int tiles_per_img = (N + TT - 1) / TT;
  const int G = gridDim.x;
  const int wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;     // images b = wg + j*G
  const int n_items = n_img * tiles_per_img;
  if (n_items <= 0) return;
  const int q0 = w * QW;
  const int nq = (Q - q0) < QW ? ((Q - q0) > 0 ? (Q - q0) : 0) : QW;   // real queries of this wave
  // byte offset of this lane's 16-byte chunk k inside a token row; lanes past the end of the row
  // (last piece, D % 256 != 0) re-read chunk 0: finite data that meets a zero query weight.
  int coff[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) coff[k] = ((lane + 64 * k) < nchunk ? (lane + 64 * k) : 0) * 16;

  // queries of this wave, pre-scaled like the reference (q = cls_token * scale, ep.py:39)
  f4 cq[QW][KP];
  auto load_cls = [&](int b) {
#pragma unroll
    for (int j = 0; j < QW; ++j)
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        int c = lane + 64 * i;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + j < Q && c < nchunk)
          v = *reinterpret_cast<const f4*>(p.cls + (int64_t)b * p.cls_bstride + (int64_t)(q0 + j) * D + 4 * c);
        cq[j][i] = v * p.scale;
      }
  };
  load_cls(wg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of ours is in flight before the ring starts

  // producer cursor
  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  auto produce = [&]() {
    if (pi < n_items) {
      const int b = wg + pimg * G;
      const int n0 = ptile * TT;
      const int rows = (N - n0) < TT ? (N - n0) : TT;
      const char* src = reinterpret_cast<const char*>(p.x + (int64_t)b * p.x_bstride + (int64_t)n0 * D);
      dma_item<NW>(src, rows * rowbytes, ring + pslot * p.slot_bytes, npiece, p.kdma, w, lane);
      ++pi;
      if (++pslot == p.nslot) pslot = 0;
      if (++ptile == tiles_per_img) { ptile = 0; ++pimg; }
    }
  };
  for (int s = 0; s < p.nslot - 1; ++s) produce();

  f4 acc[QW][KP];
  float m[QW], lsum[QW];
  int cimg = 0, ctile = 0, cslot = 0;
  int nst = 0;                                   // VMEM ops issued by this wave after its last DMA
  for (int i = 0; i < n_items; ++i) {
    // ---- wait for item i, free slot of item i-1, refill it ------------------------------
    int ahead = pi - 1 - i;                      // items issued after item i
    wait_vmcnt(ahead * p.kdma + nst);
    ring_barrier();
    {
      const int before = pi;
      produce();
      if (pi != before) nst = 0;
    }
    const int b = wg + cimg * G;
    const int n0 = ctile * TT;
    const int nvalid = (N - n0) < TT ? (N - n0) : TT;
    const char* tile = ring + cslot * p.slot_bytes;
    if (++cslot == p.nslot) cslot = 0;
    if (ctile == 0) {
      if (p.cls_bstride != 0 && cimg != 0) load_cls(b);   // per-image query override (rare path)
#pragma unroll
      for (int j = 0; j < QW; ++j) {
        m[j] = -INFINITY; lsum[j] = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[j][k] = f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // ---- compute ------------------------------------------------------------------------
#pragma unroll
    for (int t0 = 0; t0 < TT; t0 += TB) {
      if (t0 < nvalid) {
        f4 xv[TB][KP];
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
          for (int k = 0; k < KP; ++k)
            xv[t][k] = *reinterpret_cast<const f4*>(tile + (t0 + t) * rowbytes + coff[k]);
        float part[QW][TB];
#pragma unroll
        for (int j = 0; j < QW; ++j)
#pragma unroll
          for (int t = 0; t < TB; ++t) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
              s = fmaf(cq[j][k].x, xv[t][k].x, s);
              s = fmaf(cq[j][k].y, xv[t][k].y, s);
              s = fmaf(cq[j][k].z, xv[t][k].z, s);
              s = fmaf(cq[j][k].w, xv[t][k].w, s);
            }
            part[j][t] = s;
          }
        float u[QW];
        butterfly<QW>(part, u);
        const int row = lane >> 4;
        const bool rowvalid = (t0 + row) < nvalid;
        float ue[QW];
        bool need = false;
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          ue[j] = rowvalid ? u[j] : -INFINITY;
          need |= ue[j] > m[j] + LAZY_MAX_THR;
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {     // wave-uniform, rare
#pragma unroll
          for (int j = 0; j < QW; ++j) {
            float mx = fmaxf(fmaxf(readlane_f(ue[j], 0), readlane_f(ue[j], 16)),
                             fmaxf(readlane_f(ue[j], 32), readlane_f(ue[j], 48)));
            float mn = fmaxf(m[j], mx);
            float f = __builtin_amdgcn_exp2f((m[j] - mn) * LOG2E);   // m = -inf -> 0
            m[j] = mn;
            lsum[j] *= f;
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[j][k] *= f;
          }
        }
        float pr[QW];
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          pr[j] = __builtin_amdgcn_exp2f((ue[j] - m[j]) * LOG2E);    // invalid rows: exp2(-inf) = 0
          lsum[j] += pr[j];
        }
        // raw scores for backward / attention maps
        if ((lane & 15) == 0 && rowvalid) {
#pragma unroll
          for (int j = 0; j < QW; ++j)
            if (q0 + j < Q) p.S[((int64_t)b * Q + q0 + j) * N + n0 + t0 + row] = u[j];
        }
        nst += nq;   // exactly the stores issued above (lane 0 is always active in them)
#pragma unroll
        for (int j = 0; j < QW; ++j)
#pragma unroll
          for (int t = 0; t < TB; ++t) {
            const float a = readlane_f(pr[j], 16 * t);
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[j][k] += a * xv[t][k];
          }
      }
    }
    // ---- image epilogue -------------------------------------------------------------------
    if (ctile == tiles_per_img - 1) {
#pragma unroll
      for (int j = 0; j < QW; ++j) {
        const float l = readlane_f(lsum[j], 0) + readlane_f(lsum[j], 16) +
                        readlane_f(lsum[j], 32) + readlane_f(lsum[j], 48);
        const float inv = 1.0f / l;
        if (q0 + j < Q) {
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            const int c = lane + 64 * k;
            if (c < nchunk)
              *reinterpret_cast<f4*>(p.P + ((int64_t)b * Q + q0 + j) * D + 4 * c) = acc[j][k] * inv;
          }
          if (lane == 0) {
            f4 rec = {m[j], l, 0.f, 0.f};
            *reinterpret_cast<f4*>(p.ML + ((int64_t)b * Q + q0 + j) * 4) = rec;
          }
        }
      }
      nst += nq * (KP + 1);
      ctile = 0; ++cimg;
    } else {
      ++ctile;
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward (gradient of cls_token)
// ring items per image: H = ceil(Q/TT) header items holding rows of dP[b], then the token tiles.
// Every item additionally carries one 4-byte-per-lane DMA per wave into a private 256-byte area:
// header items fetch ML[b,q,0:4] of the wave's queries, token items fetch S[b,q,n0:n0+TT].
// ---------------------------------------------------------------------------------------
template <int QW, int KP, int NW, int TT>
__global__ __launch_bounds__(NW * 64, 2) void ep_pool_bwd_kernel(PoolParams p) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const int rowbytes = D * 4;
  const int nchunk = D >> 2;
  const int npiece = p.slot_bytes >> 10;
  const int tiles_per_img = (N + TT - 1) / TT;
  const int H = (Q + TT - 1) / TT;
  const int items_per_img = H + tiles_per_img;
  const int G = gridDim.x;
  const int wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * items_per_img;
  const int q0 = w * QW;
  char* small_base = ring + p.nslot * p.slot_bytes;      // [nslot][NW][64 floats]
  int coff[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) coff[k] = ((lane + 64 * k) < nchunk ? (lane + 64 * k) : 0) * 16;

  f4 gacc[QW][KP];
#pragma unroll
  for (int j = 0; j < QW; ++j)
#pragma unroll
    for (int k = 0; k < KP; ++k) gacc[j][k] = f4{0.f, 0.f, 0.f, 0.f};

  if (n_items > 0) {
    int pi = 0, pimg = 0, pidx = 0, pslot = 0;
    auto produce = [&]() {
      if (pi < n_items) {
        const int b = wg + pimg * G;
        const int slot = pslot;
        if (++pslot == p.nslot) pslot = 0;
        char* small = small_base + (slot * NW + w) * 256;
        if (pidx < H) {                                   // header: rows of dP[b]
          const int r0 = pidx * TT;
          const int rows = (Q - r0) < TT ? (Q - r0) : TT;
          const char* src = reinterpret_cast<const char*>(p.dP + ((int64_t)b * Q + r0) * D);
          dma_item<NW>(src, rows * rowbytes, ring + slot * p.slot_bytes, npiece, p.kdma, w, lane);
          // ML[b, q0 + (lane>>2), lane&3]
          int qq = q0 + (lane >> 2);
          qq = qq < Q ? qq : Q - 1;
          const float* ms = p.ML + ((int64_t)b * Q + qq) * 4 + (lane & 3);
          __builtin_amdgcn_global_load_lds((gptr_t)ms, (lds_ptr_t)small, 4, 0, 0);
        } else {
          const int n0 = (pidx - H) * TT;
          const int rows = (N - n0) < TT ? (N - n0) : TT;
          const char* src = reinterpret_cast<const char*>(p.x + (int64_t)b * p.x_bstride + (int64_t)n0 * D);
          dma_item<NW>(src, rows * rowbytes, ring + slot * p.slot_bytes, npiece, p.kdma, w, lane);
          // S[b, q0 + lane/TT, n0 + lane%TT]
          int qq = q0 + lane / TT;
          qq = qq < Q ? qq : Q - 1;
          int nn = n0 + lane % TT;
          nn = nn < N ? nn : N - 1;
          const float* ss = p.S + ((int64_t)b * Q + qq) * N + nn;
          __builtin_amdgcn_global_load_lds((gptr_t)ss, (lds_ptr_t)small, 4, 0, 0);
        }
        ++pi;
        if (++pidx == items_per_img) { pidx = 0; ++pimg; }
      }
    };
    for (int s = 0; s < p.nslot - 1; ++s) produce();

    f4 gq[QW][KP];                 // dP rows of this wave's queries for the current image
    float mq[QW], il[QW], dl[QW];  // row max, 1/l, delta of this wave's queries
    int cidx = 0, cslot = 0;
    const int kd = p.kdma + 1;
    for (int i = 0; i < n_items; ++i) {
      const int ahead = pi - 1 - i;
      wait_vmcnt(ahead * kd);
      ring_barrier();
      produce();
      const int slot = cslot;
      if (++cslot == p.nslot) cslot = 0;
      const char* tile = ring + slot * p.slot_bytes;
      const float* small = reinterpret_cast<const float*>(small_base + (slot * NW + w) * 256);
      if (cidx < H) {
        // header item: pick up the dP rows of my queries that live in this item
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          const int q = q0 + j;
          if (q < Q && q / TT == cidx) {
            const int r = q % TT;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
              f4 v = *reinterpret_cast<const f4*>(tile + r * rowbytes + coff[k]);
              if (lane + 64 * k >= nchunk) v = f4{0.f, 0.f, 0.f, 0.f};
              gq[j][k] = v;
            }
          } else if (q >= Q && cidx == 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) gq[j][k] = f4{0.f, 0.f, 0.f, 0.f};
          }
          mq[j] = small[4 * j + 0];
          il[j] = 1.0f / small[4 * j + 1];
          dl[j] = small[4 * j + 2];
        }
      } else {
        const int n0 = (cidx - H) * TT;
        const int nvalid = (N - n0) < TT ? (N - n0) : TT;
#pragma unroll
        for (int t0 = 0; t0 < TT; t0 += TB) {
          if (t0 < nvalid) {
            f4 xv[TB][KP];
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
              for (int k = 0; k < KP; ++k)
                xv[t][k] = *reinterpret_cast<const f4*>(tile + (t0 + t) * rowbytes + coff[k]);
            float part[QW][TB];
#pragma unroll
            for (int j = 0; j < QW; ++j)
#pragma unroll
              for (int t = 0; t < TB; ++t) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                  s = fmaf(gq[j][k].x, xv[t][k].x, s);
                  s = fmaf(gq[j][k].y, xv[t][k].y, s);
                  s = fmaf(gq[j][k].z, xv[t][k].z, s);
                  s = fmaf(gq[j][k].w, xv[t][k].w, s);
                }
                part[j][t] = s;
              }
            float u[QW];
            butterfly<QW>(part, u);                       // dA[q][t] in row t
            const int row = lane >> 4;
            const bool rowvalid = (t0 + row) < nvalid;
            float wgt[QW];
#pragma unroll
            for (int j = 0; j < QW; ++j) {
              const float s = small[j * TT + t0 + row];
              const float a = __builtin_amdgcn_exp2f((s - mq[j]) * LOG2E) * il[j];
              wgt[j] = rowvalid ? a * (u[j] - dl[j]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < QW; ++j)
#pragma unroll
              for (int t = 0; t < TB; ++t) {
                const float a = readlane_f(wgt[j], 16 * t);
#pragma unroll
                for (int k = 0; k < KP; ++k) gacc[j][k] += a * xv[t][k];
              }
          }
        }
      }
      if (++cidx == items_per_img) cidx = 0;
    }
  }
  // per-workgroup partial of sum_b sum_n dS x  (reduced + scaled by ep_reduce_partials)
#pragma unroll
  for (int j = 0; j < QW; ++j)
    if (q0 + j < Q) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int c = lane + 64 * k;
        if (c < nchunk)
          *reinterpret_cast<f4*>(p.Gpart + ((int64_t)wg * Q + q0 + j) * D + 4 * c) = gacc[j][k];
      }
    }
}

// out[j] = (accumulate ? out[j] : 0) + alpha * sum_i part[i*n + j]   (deterministic order)
__global__ void ep_reduce_partials_kernel(const float* __restrict__ part, int nparts, int n,
                                          float alpha, int accumulate, float* __restrict__ out) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (j >= n) return;
  f4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0};
  int i = 0;
  for (; i + 4 <= nparts; i += 4) {
    s0 += *reinterpret_cast<const f4*>(part + (int64_t)(i + 0) * n + j);
    s1 += *reinterpret_cast<const f4*>(part + (int64_t)(i + 1) * n + j);
    s2 += *reinterpret_cast<const f4*>(part + (int64_t)(i + 2) * n + j);
    s3 += *reinterpret_cast<const f4*>(part + (int64_t)(i + 3) * n + j);
  }
  for (; i < nparts; ++i) s0 += *reinterpret_cast<const f4*>(part + (int64_t)i * n + j);
  f4 s = ((s0 + s1) + (s2 + s3)) * alpha;
  if (accumulate) s += *reinterpret_cast<const f4*>(out + j);
  *reinterpret_cast<f4*>(out + j) = s;
}

// ---------------------------------------------------------------------------------------
// generic fallback (any Q, any D % 4 == 0): one workgroup per image, two passes over the image
// (the second one is served by L2 / Infinity Cache).  Used for shapes the streaming kernel
// does not cover (D % 64 != 0, D > 1536, Q > 32) and as an in-library cross-check.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ep_pool_fwd_generic_kernel(PoolParams p) {
  extern __shared__ float sm[];       // scores of one query row: N floats, then 8 scratch
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const float* xb = p.x + (int64_t)b * p.x_bstride;
  float* red = sm + N;
  for (int q = 0; q < Q; ++q) {
    const float* cq = p.cls + (int64_t)b * p.cls_bstride + (int64_t)q * D;
    for (int n = w; n < N; n += 4) {
      float s = 0.f;
      for (int d = lane * 4; d < D; d += 256) {
        f4 xv = *reinterpret_cast<const f4*>(xb + (int64_t)n * D + d);
        f4 cv = *reinterpret_cast<const f4*>(cq + d) * p.scale;
        s = fmaf(cv.x, xv.x, s); s = fmaf(cv.y, xv.y, s); s = fmaf(cv.z, xv.z, s); s = fmaf(cv.w, xv.w, s);
      }
      s = wave_sum(s);
      if (lane == 0) { sm[n] = s; p.S[((int64_t)b * Q + q) * N + n] = s; }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int n = threadIdx.x; n < N; n += 256) mx = fmaxf(mx, sm[n]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float l = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
      float e = __builtin_amdgcn_exp2f((sm[n] - mx) * LOG2E);
      sm[n] = e; l += e;
    }
    l = wave_sum(l);
    if (lane == 0) red[w] = l;
    __syncthreads();
    l = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / l;
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
      f4 a = {0, 0, 0, 0};
      for (int n = 0; n < N; ++n) a += sm[n] * *reinterpret_cast<const f4*>(xb + (int64_t)n * D + d);
      *reinterpret_cast<f4*>(p.P + ((int64_t)b * Q + q) * D + d) = a * inv;
    }
    if (threadIdx.x == 0) {
      f4 rec = {mx, l, 0.f, 0.f};
      *reinterpret_cast<f4*>(p.ML + ((int64_t)b * Q + q) * 4) = rec;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void ep_pool_bwd_generic_kernel(PoolParams p) {
  extern __shared__ float sm[];       // weights of one query row: N floats
  const int b = blockIdx.x;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = p.D, N = p.N, Q = p.Q;
  const float* xb = p.x + (int64_t)b * p.x_bstride;
  for (int q = 0; q < Q; ++q) {
    const float* g = p.dP + ((int64_t)b * Q + q) * D;
    const float* ml = p.ML + ((int64_t)b * Q + q) * 4;
    const float mx = ml[0], inv = 1.0f / ml[1], delta = ml[2];
    for (int n = w; n < N; n += 4) {
      float s = 0.f;
      for (int d = lane * 4; d < D; d += 256) {
        f4 xv = *reinterpret_cast<const f4*>(xb + (int64_t)n * D + d);
        f4 gv = *reinterpret_cast<const f4*>(g + d);
        s = fmaf(gv.x, xv.x, s); s = fmaf(gv.y, xv.y, s); s = fmaf(gv.z, xv.z, s); s = fmaf(gv.w, xv.w, s);
      }
      s = wave_sum(s);
      if (lane == 0) {
        const float a = __builtin_amdgcn_exp2f((p.S[((int64_t)b * Q + q) * N + n] - mx) * LOG2E) * inv;
        sm[n] = a * (s - delta);
      }
    }
    __syncthreads();
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
      f4 a = {0, 0, 0, 0};
      for (int n = 0; n < N; ++n) a += sm[n] * *reinterpret_cast<const f4*>(xb + (int64_t)n * D + d);
      *reinterpret_cast<f4*>(p.Gpart + ((int64_t)b * Q + q) * D + d) = a;
    }
    __syncthreads();
  }
}

__global__ void ep_attention_kernel(const float* __restrict__ S, const float* __restrict__ ML,
                                    int rows, int N, float* __restrict__ A) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const float mx = ML[(int64_t)r * 4], inv = 1.0f / ML[(int64_t)r * 4 + 1];
  for (int n = threadIdx.x; n < N; n += blockDim.x)
    A[(int64_t)r * N + n] = __builtin_amdgcn_exp2f((S[(int64_t)r * N + n] - mx) * LOG2E) * inv;
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
struct StreamCfg {
  int qw, kp, nw, tt;
  int nslot, slot_bytes, kdma, grid;
  size_t lds_bytes;
  bool ok;
};

static StreamCfg pick_stream_cfg(int B, int N, int D, int Q, bool bwd) {
  StreamCfg c{};
  c.ok = false;
  (void)N;
  if (D % 64 != 0 || D > 1536 || Q < 1 || Q > 32) return c;
  c.kp = (D + 255) / 256;
  // queries per wave (QW) x waves per workgroup (NW) >= Q; 8 waves per CU (2 per SIMD)
  if (Q == 1) { c.qw = 1; c.nw = 1; }
  else if (Q == 2) { c.qw = 1; c.nw = 2; }
  else if (Q <= 4) { c.qw = 1; c.nw = 4; }
  else if (Q <= 8) { c.qw = 2; c.nw = 4; }
  else if (Q <= 16) { c.qw = 2; c.nw = 8; }
  else { c.qw = 4; c.nw = 8; }
  if (c.qw == 4 && c.kp > 3) return c;            // register budget: 2*QW*KP*4 accumulator VGPRs
  c.tt = 4;
  c.slot_bytes = c.tt * D * 4;
  const int npiece = c.slot_bytes / 1024;
  c.kdma = (npiece + c.nw - 1) / c.nw;
  const int wg_per_cu = 8 / c.nw;
  const size_t budget = (160 * 1024) / wg_per_cu;
  const size_t small = bwd ? (size_t)c.nw * 256 : 0;
  c.nslot = (int)(budget / (c.slot_bytes + small));
  if (c.nslot > 8) c.nslot = 8;
  if (c.nslot < 3) return c;
  // keep the counted waits inside the vmcnt range handled by wait_vmcnt()
  while ((c.nslot - 2) * (c.kdma + 1) + c.qw * (c.kp + 2) > 48 && c.nslot > 3) --c.nslot;
  c.lds_bytes = (size_t)c.nslot * (c.slot_bytes + small);
  int grid = cu_count() * wg_per_cu;
  if (grid > B) grid = B;
  c.grid = grid;
  c.ok = true;
  return c;
}

template <int QW, int KP, int NW>
static int launch_stream(bool bwd, const PoolParams& p, const StreamCfg& c, hipStream_t st) {
  constexpr int TT = 4;
  auto kf = ep_pool_fwd_kernel<QW, KP, NW, TT>;
  auto kb = ep_pool_bwd_kernel<QW, KP, NW, TT>;
  const void* fn = bwd ? (const void*)kb : (const void*)kf;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", c.lds_bytes, hipGetErrorString(e)); return (int)e; }
  if (bwd) hipLaunchKernelGGL(kb, dim3(c.grid), dim3(NW * 64), c.lds_bytes, st, p);
  else hipLaunchKernelGGL(kf, dim3(c.grid), dim3(NW * 64), c.lds_bytes, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_bwd_kernel" : "ep_pool_fwd_kernel");
  return 0;
}

template <int QW, int NW>
static int dispatch_kp(bool bwd, const PoolParams& p, const StreamCfg& c, hipStream_t st) {
  switch (c.kp) {
    case 1: return launch_stream<QW, 1, NW>(bwd, p, c, st);
    case 2: return launch_stream<QW, 2, NW>(bwd, p, c, st);
    case 3: return launch_stream<QW, 3, NW>(bwd, p, c, st);
    case 4: return launch_stream<QW, 4, NW>(bwd, p, c, st);
    case 5: if constexpr (QW <= 2) return launch_stream<QW, 5, NW>(bwd, p, c, st); break;
    case 6: if constexpr (QW <= 2) return launch_stream<QW, 6, NW>(bwd, p, c, st); break;
  }
  set_error("no streaming kernel for qw=%d kp=%d", QW, c.kp);
  return EP_E_UNSUPPORTED;
}

static int dispatch_stream(bool bwd, const PoolParams& p, const StreamCfg& c, hipStream_t st) {
  if (c.qw == 1 && c.nw == 1) return dispatch_kp<1, 1>(bwd, p, c, st);
  if (c.qw == 1 && c.nw == 2) return dispatch_kp<1, 2>(bwd, p, c, st);
  if (c.qw == 1 && c.nw == 4) return dispatch_kp<1, 4>(bwd, p, c, st);
  if (c.qw == 2 && c.nw == 4) return dispatch_kp<2, 4>(bwd, p, c, st);
  if (c.qw == 2 && c.nw == 8) return dispatch_kp<2, 8>(bwd, p, c, st);
  if (c.qw == 4 && c.nw == 8) return dispatch_kp<4, 8>(bwd, p, c, st);
  set_error("no streaming kernel for qw=%d nw=%d", c.qw, c.nw);
  return EP_E_UNSUPPORTED;
}

static int g_force_generic = -1;
int debug_force_generic(int on) { int old = g_force_generic == 1; g_force_generic = on ? 1 : 0; return old; }
static bool force_generic() {
  if (g_force_generic < 0) {
    const char* e = getenv("EP_POOL_FORCE_GENERIC");
    g_force_generic = (e && e[0] == '1') ? 1 : 0;
  }
  return g_force_generic == 1;
}

size_t pool_workspace_bytes(int B, int N, int D, int Q) {
  // Gpart: one (Q,D) partial per workgroup of the backward; the generic kernel uses one per
  // image, the streaming kernel one per resident workgroup (<= B).
  (void)N;
  return round_up((size_t)B * Q * D * sizeof(float), 256);
}

int pool_forward(const PoolParams& p0, hipStream_t st) {
  PoolParams p = p0;
  StreamCfg c = pick_stream_cfg(p.B, p.N, p.D, p.Q, false);
  if (c.ok && !force_generic()) {
    p.nslot = c.nslot; p.slot_bytes = c.slot_bytes; p.kdma = c.kdma;
    return dispatch_stream(false, p, c, st);
  }
  const size_t lds = (size_t)(p.N + 8) * sizeof(float);
  hipLaunchKernelGGL(ep_pool_fwd_generic_kernel, dim3(p.B), dim3(256), lds, st, p);
  EP_LAUNCH_CHECK("ep_pool_fwd_generic_kernel");
  return 0;
}

int pool_backward(const PoolParams& p0, float* dcls, int accumulate, hipStream_t st) {
  PoolParams p = p0;
  StreamCfg c = pick_stream_cfg(p.B, p.N, p.D, p.Q, true);
  int nparts;
  if (c.ok && !force_generic()) {
    p.nslot = c.nslot; p.slot_bytes = c.slot_bytes; p.kdma = c.kdma;
    EP_TRY(dispatch_stream(true, p, c, st));
    nparts = c.grid;
  } else {
    const size_t lds = (size_t)(p.N + 8) * sizeof(float);
    hipLaunchKernelGGL(ep_pool_bwd_generic_kernel, dim3(p.B), dim3(256), lds, st, p);
    EP_LAUNCH_CHECK("ep_pool_bwd_generic_kernel");
    nparts = p.B;
  }
  const int n = p.Q * p.D;
  hipLaunchKernelGGL(ep_reduce_partials_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, st,
                     p.Gpart, nparts, n, p.scale, accumulate, dcls);
  EP_LAUNCH_CHECK("ep_reduce_partials_kernel");
  return 0;
}

int attention_from_scores(const float* S, const float* ML, int rows, int N, float* A, hipStream_t st) {
  hipLaunchKernelGGL(ep_attention_kernel, dim3(rows), dim3(256), 0, st, S, ML, rows, N, A);
  EP_LAUNCH_CHECK("ep_attention_kernel");
  return 0;
}

}  // namespace ep
