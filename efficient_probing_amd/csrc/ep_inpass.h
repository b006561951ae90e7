// In-pass contractions of the fused EP step (gfx950): the two projections around the BatchNorm/classifier chain are
// computed INSIDE the token passes by the pooling workgroups themselves, so that they cost the step (almost) no time of
// their own -- the matrix pipe is idle while the passes stream, and HBM is idle while the contractions run:
//
//   first pass  : y[b, q Dq + c] = P[b, q, :] . Wv[q Dq + c, :]   (reference poolings/ep.py:40 after pool-then-project)
//                 as four K-quarter partials, by workgroups that have finished their images while the others stream;
//   second pass : dP[b, q, :] = dy[b, q-slice] Wv[q-slice, :]      (autograd of the same line)
//                 by every pooling workgroup in front of its own stream (no contraction launch before the pass).
//
// Work unit: images are grouped into row blocks of 32; a row block needs 32 tasks = 8 queries x 4 quarters (K quarters
// for y, column quarters for dP), and image b of the batch names task b: (row block b / 32, query (b % 32) / 4,
// quarter b % 4).  Hand-off between workgroups (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md
// Guideline 16): the handed-off rows are stored WRITE-THROUGH (16-byte `sc1` buffer stores, every 128-byte line written
// whole by one instruction), every storing wave drains its stores (s_waitcnt vmcnt(0)) and adds for itself to the row
// block's arrival counter (agent-scope atomic; zero at launch); a consumer polls the counter from one lane (sc1 load +
// s_sleep), passes a workgroup barrier and then reads the rows with `sc1` loads -- buffer loads to registers (first
// pass) or `sc1` LDS-DMA copies (the header items of the second pass) -- so no acquire either.  No release fences: one
// `buffer_wbl2` per wave and image measured +220 us per pass (4096 L2 write-backs).
//
// Progress: a producer never waits before it has arrived (tasks first, waits after).  First-round row blocks are produced
// by the consumer's own aligned group of 32 workgroups (dispatch is in block order, so they are resident with it); the
// row blocks of later rounds are produced by the workgroups that stream the fewest images -- higher block indices, which
// become resident at the latest when single-image workgroups in front of them retire (those wait only on their own
// group).  So the launch makes progress whenever more workgroups are resident than stream an image of the last round
// (768 against 256 at B = 1024); and every wait is bounded all the same (PoolParams.ip_err counts give-ups; tests
// assert it stays zero).
//
// Shapes: Q = 8, D = 256 KT (KT = 1..3), d_out = 1, B % 32 == 0, pooling grid % 32 == 0 (host-checked: ep_pool.hip).
#pragma once
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_side.h"

namespace ep {

constexpr int IP_WAVES = 4;                      // waves per workgroup of the kernels that carry these tasks
constexpr int IP_TARGET = 32 * IP_WAVES;         // arrivals that complete a row block
constexpr int IP_SPIN_LIMIT = 1 << 19;           // x ~1.3 us per poll: gives up after ~0.7 s instead of hanging the GPU
// One counter per 128-byte line, polled about once a microsecond.  With the 32 counters of a 1024-image batch on ONE line
// and 0.3 us polls, the few hundred polling workgroups of a pass (sc1 loads: every poll crosses the fabric to the line's
// home) cut the token stream of everybody else to 40 % (measured with EP_IP_STAMP: 2.2 TB/s while pollers were active).
constexpr int IP_CNT_STRIDE = 32;                // ints between the counters of consecutive row blocks

// raw buffer view of a handed-off matrix: 16-byte loads / stores with the sc1 bit (aux 16) that the compiler still
// tracks in its vmcnt bookkeeping (inline-asm stores would not be)
constexpr int IP_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ip_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
}
__device__ __forceinline__ void ip_store16_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f4v v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, IP_SC1);
}
__device__ __forceinline__ f4v ip_load16_coherent(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, IP_SC1);
}

// this wave's write-through stores have left the CU (drain), then ONE arrival for the wave
__device__ __forceinline__ void ip_arrive(int* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane_id() == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup: returns once *cnt >= target, polled by one lane; the barrier stands between the poll and every load
// of the handed-off rows.  ACQUIRE: additionally ONE agent-scope acquire (this CU's L1) completed in front of the
// barrier, for readers that are not sc1 loads to registers (the LDS-DMA ring).
template <bool ACQUIRE>
__device__ __forceinline__ void ip_wait(int* cnt, int target, int* err) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > IP_SPIN_LIMIT) { if (err) atomicAdd(err, 1); break; }
      __builtin_amdgcn_s_sleep(32);
    }
    if (ACQUIRE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the invalidate has completed before the barrier opens
    }
  }
  __syncthreads();
}

// LDS bytes the tasks need (the token ring of the pass is idle while they run)
constexpr size_t ip_dp_lds_bytes(int kt) {      // A image + one B tile (>= the 32 x 68 output staging tile that reuses it) + 3 x 32 kt column scalars
  return (size_t)(32 * (((32 * kt + 29) / 64) * 64 + 34) + (32 * kt * (64 + 16) > 32 * 68 ? 32 * kt * (64 + 16) : 32 * 68) + 3 * 32 * kt) * sizeof(float);
}
// BatchNorm backward folded into the dP tasks (PoolParams.ip_fold_*): dz, z (B x D), rstd (D), the per-32-row-tile column
// statistics of dz (cs: [B / 32][2][D]) and where column quarter 0 writes dy
struct IpFold { const float* dz; const float* z; const float* rstd; const float* cs; float* dy; };
constexpr size_t ip_y_lds_bytes(int kt) { return (size_t)2 * (32 + 32 * kt) * (BK + 2) * sizeof(float); }

// ---- second pass: dP rows of one (row block, query, column quarter) --------------------------------------------
// A = dy[32 rows, q-slice of DQ = 32 KT columns] (K layout), B = Wv[q-slice rows, column quarter of 64 KT] (T layout: k-rows
// of contiguous output columns).  Everything is fetched up front (KT + 2 KT KT float4 per thread), one round trip; the
// A image stays in LDS, the KT 64-column B tiles pass through one LDS buffer.  Exact fp32 (v_mfma_f32_16x16x4_f32).
template <int KT>
__device__ __forceinline__ void ip_dp_task_body(const float* __restrict__ dy, const float* __restrict__ Wv, float* dP, int nB, int b,
                                               char* lds_raw, const IpFold fold = IpFold{nullptr, nullptr, nullptr, nullptr, nullptr}) {
  constexpr int DQ = 32 * KT, D = 256 * KT, Q = 8, CW = 64 * KT;
  constexpr int SA = ((DQ + 29) / 64) * 64 + 34;      // A row stride in floats, == 34 (mod 64): conflict-free fragment reads
  constexpr int SB = 64 + 16;                         // B k-row stride
  float* As = reinterpret_cast<float*>(lds_raw);      // [32][SA]
  float* Bs = As + 32 * SA;                           // [DQ][SB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id_uniform();
  const int wm = w >> 1, wn = w & 1, i16 = lane & 15, kk = lane >> 4;
  const int rb = b >> 5, r = b & 31, q = r >> 2, ch = r & 3;
  const int row0 = rb * 32;
  const float* A = dy + (int64_t)row0 * D + q * DQ;
  const float* W = Wv + (int64_t)q * DQ * D + ch * CW;
  constexpr int SC = 64 + 4;                          // row stride of the output staging tile (it reuses the B buffer)
  float* Cs = Bs;
  const __amdgpu_buffer_rsrc_t rC = ip_rsrc(dP, (size_t)nB * Q * D * sizeof(float));
  const unsigned c_off = (unsigned)((((int64_t)row0 * Q + q) * D + ch * CW) * sizeof(float));

  f4v ra[KT], rz[KT], rw[KT][2 * KT];
  float* cst = Bs + (DQ * SB > 32 * 68 ? DQ * SB : 32 * 68);   // [m1 | m2 | rstd][DQ] column scalars of the folded BatchNorm backward
  if (fold.dz) {
    // A = dy is formed HERE: dy = rstd (dz - m1 - z m2), m1 = mean_b dz, m2 = mean_b dz z (probe_heads.py:109-110
    // BatchNorm1d(affine=False) backward); the means from the per-tile column sums the dz contraction left, summed in tile order
    const float* Az = fold.dz + (int64_t)row0 * D + q * DQ;
    const float* Zz = fold.z + (int64_t)row0 * D + q * DQ;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      const int idx = tid + 256 * i;
      const int64_t at = (int64_t)(idx / (DQ / 4)) * D + 4 * (idx % (DQ / 4));
      ra[i] = *reinterpret_cast<const f4v*>(Az + at);
      rz[i] = *reinterpret_cast<const f4v*>(Zz + at);
    }
  } else {
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      const int idx = tid + 256 * i;
      ra[i] = *reinterpret_cast<const f4v*>(A + (int64_t)(idx / (DQ / 4)) * D + 4 * (idx % (DQ / 4)));
    }
  }
#pragma unroll
  for (int nt = 0; nt < KT; ++nt)
#pragma unroll
    for (int i = 0; i < 2 * KT; ++i) {
      const int idx = tid + 256 * i;
      rw[nt][i] = *reinterpret_cast<const f4v*>(W + (int64_t)(idx >> 4) * D + nt * 64 + 4 * (idx & 15));
    }
  if (fold.dz) {
    const int ntile = nB >> 5;
    for (int t = tid; t < 3 * DQ; t += 256) {
      if (t < 2 * DQ) {                                // column sums over the batch: 32-row tiles in order
        const int which = t / DQ, col = t - which * DQ;
        const float* c = fold.cs + (int64_t)which * D + q * DQ + col;
        float acc_ = 0.f;
        for (int k0 = 0; k0 < ntile; k0 += 16) {       // 16 loads in flight per trip (clamped address, masked value)
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = c[(int64_t)((k0 + u) < ntile ? (k0 + u) : ntile - 1) * 2 * D];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc_ += (k0 + u) < ntile ? v[u] : 0.f;
        }
        cst[t] = acc_ / (float)nB;
      } else {
        cst[t] = fold.rstd[q * DQ + t - 2 * DQ];
      }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rY = ip_rsrc(fold.dy, (size_t)nB * D * sizeof(float));
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      const int idx = tid + 256 * i, row = idx / (DQ / 4), c0 = 4 * (idx % (DQ / 4));
      f4v v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = cst[2 * DQ + c0 + j] * (ra[i][j] - cst[c0 + j] - rz[i][j] * cst[DQ + c0 + j]);
      ra[i] = v;
      // dy itself is read by the delta items of the pooling workgroups and by the dWv side tasks of this launch: the
      // tasks of column quarter 0 publish it (write-through, whole 128-byte lines per instruction)
      if (ch == 0) ip_store16_wt(rY, (unsigned)((((int64_t)(row0 + row)) * D + q * DQ + c0) * sizeof(float)), v);
    }
  }
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    const int idx = tid + 256 * i;
    float* d = As + (idx / (DQ / 4)) * SA + 4 * (idx % (DQ / 4));
    *reinterpret_cast<f2*>(d) = f2{ra[i].x, ra[i].y};
    *reinterpret_cast<f2*>(d + 2) = f2{ra[i].z, ra[i].w};
  }
#pragma unroll
  for (int nt = 0; nt < KT; ++nt) {
    if (nt > 0) __syncthreads();                       // every wave is done reading the previous output tile
#pragma unroll
    for (int i = 0; i < 2 * KT; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<f4v*>(Bs + (idx >> 4) * SB + 4 * (idx & 15)) = rw[nt][i];
    }
    __syncthreads();
    f4v acc[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float af[8], bf[8][2];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = 32 * kt + 4 * s + kk;
        af[s] = As[(wm * 16 + i16) * SA + k];
        bf[s][0] = Bs[k * SB + wn * 32 + i16];
        bf[s][1] = Bs[k * SB + wn * 32 + 16 + i16];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s][1], acc[1], 0, 0, 0);
      }
    }
    // The rows are handed to other workgroups inside this launch: 16-byte write-through stores, every 128-byte line whole
    // in one instruction -- so the tile goes through LDS once (D layout of 16x16x4: column = lane & 15, row = 4 (lane >> 4) + r)
    __syncthreads();                                   // every wave is done reading the B tile
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Cs[(wm * 16 + kk * 4 + rr) * SC + wn * 32 + ni * 16 + i16] = acc[ni][rr];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 4, c4 = idx & 15;
      const f4v v = *reinterpret_cast<const f4v*>(Cs + row * SC + 4 * c4);
      ip_store16_wt(rC, c_off + (unsigned)(((int64_t)row * Q * D + nt * 64 + 4 * c4) * sizeof(float)), v);
    }
  }
  __syncthreads();                                     // LDS free for the next task / the token ring
}

template <int KT>
__device__ __forceinline__ void ip_dp_task(const PoolParams& p, int b, char* lds_raw) {
  ip_dp_task_body<KT>(p.ip_dy, p.ip_Wv, const_cast<float*>(p.dP), p.B, b, lds_raw,
                      IpFold{p.ip_fold_dz, p.ip_fold_z, p.ip_fold_rstd, p.ip_fold_cs, const_cast<float*>(p.ip_dy)});
}
// the same as a CALL: inside the ticketed second pass the task sits in the middle of the token loop, and inlined there its
// ~90 staging registers pushed the loop's own values into scratch (64 spilled registers); as a function the caller only
// saves what is live across the (rare) call
template <int KT>
__device__ __attribute__((noinline)) void ip_dp_task_call(const float* dy, const float* Wv, float* dP, int nB, int b, char* lds_raw) {
  ip_dp_task_body<KT>(dy, Wv, dP, nB, b, lds_raw);
}

// ---- first pass: K quarters ks0 .. ks0 + nks - 1 of y for (row block, query) -----------------------------------------
// A = P[32 rows, q, K quarter of 64 KT] (K layout, row stride Q D), B = Wv[q-slice rows (32 KT outputs), K quarter]
// (K layout).  Output 32 x 32 KT at `out` (row stride D: a K-quarter partial buffer, or y itself when all four quarters
// are summed here); wave (wm, wn) owns 16 rows x 16 KT columns.  The 2 KT K-tiles of a quarter are fetched up front
// (2 KT (1 + KT) float4 per thread), then pass through a double LDS buffer; quarters follow one another.
template <int KT>
__device__ __forceinline__ void ip_y_task(const PoolParams& p, int rb, int q, int ks0, int nks, float* out, char* lds_raw) {
  constexpr int DQ = 32 * KT, D = 256 * KT, Q = 8, NKT = 2 * KT;
  constexpr int LDK2 = BK + 2;
  constexpr int STAGE = (32 + DQ) * LDK2;             // floats per stage: A image then B image
  float* stage = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id_uniform();
  const int wm = w >> 1, wn = w & 1, i16 = lane & 15, kk = lane >> 4;
  const int row0 = rb * 32;
  const __amdgpu_buffer_rsrc_t rP = ip_rsrc(p.P, (size_t)p.B * Q * D * sizeof(float));   // rows handed over in this launch
  float* C = out + (int64_t)row0 * D + q * DQ;
  const int lrow = tid >> 3, lc = 4 * (tid & 7);
  f4v acc[KT];
#pragma unroll
  for (int bi = 0; bi < KT; ++bi) acc[bi] = f4v{0.f, 0.f, 0.f, 0.f};
  for (int ks = ks0; ks < ks0 + nks; ++ks) {
    const unsigned a_off = (unsigned)((((int64_t)row0 * Q + q) * D + ks * (64 * KT)) * sizeof(float));
    const float* W = p.ip_WvF + (int64_t)q * DQ * D + ks * (64 * KT);
    f4v ra[NKT], rw[NKT][KT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      ra[t] = ip_load16_coherent(rP, a_off + (unsigned)(((int64_t)lrow * Q * D + 32 * t + lc) * sizeof(float)));
#pragma unroll
      for (int i = 0; i < KT; ++i) rw[t][i] = *reinterpret_cast<const f4v*>(W + (int64_t)(lrow + 32 * i) * D + 32 * t + lc);
    }
    auto lstore = [&](int t) {
      float* sa = stage + (t & 1) * STAGE;
      float* d = sa + lrow * LDK2 + lc;
      *reinterpret_cast<f2*>(d) = f2{ra[t].x, ra[t].y};
      *reinterpret_cast<f2*>(d + 2) = f2{ra[t].z, ra[t].w};
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        float* e = sa + (32 + lrow + 32 * i) * LDK2 + lc;
        *reinterpret_cast<f2*>(e) = f2{rw[t][i].x, rw[t][i].y};
        *reinterpret_cast<f2*>(e + 2) = f2{rw[t][i].z, rw[t][i].w};
      }
    };
    lstore(0);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      if (t + 1 < NKT) lstore(t + 1);                    // the other buffer: everyone left it before the last barrier
      const float* sa = stage + (t & 1) * STAGE;
      const float* sb = sa + 32 * LDK2;
      float af[8], bf[8][KT];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        af[s] = sa[(wm * 16 + i16) * LDK2 + 4 * s + kk];
#pragma unroll
        for (int bi = 0; bi < KT; ++bi) bf[s][bi] = sb[(wn * 16 * KT + bi * 16 + i16) * LDK2 + 4 * s + kk];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int bi = 0; bi < KT; ++bi) acc[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s][bi], acc[bi], 0, 0, 0);
      __syncthreads();
    }
  }
#pragma unroll
  for (int bi = 0; bi < KT; ++bi)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      C[(int64_t)(wm * 16 + kk * 4 + rr) * D + wn * 16 * KT + bi * 16 + i16] = acc[bi][rr];
}

}  // namespace ep
