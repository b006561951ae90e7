// EP attentive pooling: the two streaming passes over the frozen tokens (gfx950 / CDNA4).
//
//   forward  (reference poolings/ep.py:35-44):  S = (cls*scale) x^T ; A = softmax_n S ; P = A x
//   backward (autograd of the same lines)     :  dA = dP x^T ; dS = A (dA - delta) ;
//                                                dcls = scale * sum_b dS x
//
// Design (DESIGN.md "pool kernels"):
//   * one workgroup streams whole images; the image's tokens are copied HBM -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of NSLOT tiles of 4
//     tokens, several tiles ahead of the compute, so x is read from HBM exactly once per pass
//     and never touches a VGPR on the way in;
//   * the Q queries are split over the NW waves of the workgroup (QW queries per wave); every
//     wave reads every token row from LDS (lane = 16-byte chunk of the row, conflict-free
//     ds_read_b128) and owns the FULL D-dimension for its queries, so there is no cross-wave
//     reduction at all -- the only synchronisation is one s_barrier per tile for the ring;
//   * scores: lane-local packed FMAs (v_pk_fma_f32), then a v_permlane32_swap /
//     v_permlane16_swap / DPP butterfly that leaves score (q, t) replicated in the 16 lanes of row
//     t of register q;
//   * softmax: lazy-max online softmax evaluated lane-parallel (row = token), weights broadcast
//     to SGPRs with v_readlane and used as the scalar operand of the pooling FMAs;
//   * LDS-DMA completion is tracked with a counted s_waitcnt vmcnt(N) whose N is a compile-time
//     constant in steady state (never 0 inside the stream).
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"
#include "ep_side.h"
#include "ep_inpass.h"
#include "ep_sidetask.h"
#include "ep_stream_dev.h"

namespace ep {

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// BF16: tokens are bf16 in memory; a ring tile then holds twice as many tokens in the same bytes and the rows are
// widened to fp32 as they are read from LDS -- everything after that is the fp32 code path unchanged.
// LN: LayerNorm-of-tokens mode (PoolParams.tokstat): every ring item carries one extra 4-byte-per-lane DMA with the
// {mean, rstd} pairs of its tokens; scores become rstd (q.x - mean sum(q)), the pooling weights a * rstd, and the
// mean term sum a rstd mean is carried as one scalar per query and subtracted at the end of the image.
template <int QW, int KP, int NW, int DFIX, bool BF16, bool LN>
__global__ __launch_bounds__(NW * 64, (stream_waves_per_cu(QW, KP, NW) / 4)) void ep_pool_fwd_kernel(PoolParams p) {
  using Cfg = StreamCfgT<QW, KP, NW>;
  constexpr int NSLOT = LN ? Cfg::NSLOT_B : Cfg::NSLOT_F, KDMA = Cfg::KDMA, TT = Cfg::TT * (BF16 ? 2 : 1);
  constexpr int KD = KDMA + (LN ? 1 : 0);          // VMEM operations per ring item
  constexpr int ES = BF16 ? 2 : 4;                 // bytes per stored token element
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int D = DFIX ? DFIX : p.D;
  const int N = p.N, Q = p.Q;
  const int rowbytes = D * ES;
  const int nchunk = D >> 2;                    // 4-element chunks per row (16 bytes of fp32 / 8 bytes of bf16)
  const int slot_bytes = TT * rowbytes;
  const int npiece = slot_bytes >> 10;
  const int tiles_per_img = (N + TT - 1) / TT;
  const int G = gridDim.x;
  const int wg = blockIdx.x;
  const int n_img = (p.B - wg + G - 1) / G;     // images b = wg + j*G
  const int n_items = n_img * tiles_per_img;
  // in-pass contractions (ep_inpass.h): this launch clears the arrival counters of the OTHER pass
  constexpr bool IPOK = (NW == 4 && QW == 2 && KP <= 3 && !LN);
  if constexpr (IPOK) {
    if (wg == 0 && p.ip_zero)
      for (int t = threadIdx.x; t < p.ip_nzero; t += NW * 64) p.ip_zero[t] = 0;
  }
  if (n_items <= 0) return;
  auto stamp = [&](int k) {                                  // diagnostic only (EP_IP_STAMP): 100 MHz timeline per workgroup
    if constexpr (IPOK) { if (p.dbg && threadIdx.x == 0) p.dbg[(int64_t)wg * 8 + k] = __builtin_amdgcn_s_memrealtime(); }
  };
  stamp(0);
  const int q0 = w * QW;
  const unsigned lane16 = (unsigned)lane * 16u;
  // byte offset of this lane's 16-byte chunk k inside a token row; lanes past the end of the row
  // (last piece, D % 256 != 0) re-read chunk 0: finite data that meets a zero query weight.
  int coff[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) coff[k] = ((lane + 64 * k) < nchunk ? (lane + 64 * k) : 0) * (4 * ES);

  // queries of this wave, pre-scaled like the reference (q = cls_token * scale, ep.py:39)
  f4 cq[QW][KP];
  auto load_cls = [&](int b) {
#pragma unroll
    for (int j = 0; j < QW; ++j)
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const int c = lane + 64 * i;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + j < Q && c < nchunk)
          v = *reinterpret_cast<const f4*>(p.cls + (int64_t)b * p.cls_bstride + (int64_t)(q0 + j) * D + 4 * c);
        cq[j][i] = v * p.scale;
      }
  };
  load_cls(wg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of ours is in flight before the ring starts
  float wsum[QW];                                    // LN: sum over D of the scaled query
  auto sum_cls = [&]() {
#pragma unroll
    for (int j = 0; j < QW; ++j) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < KP; ++k) t += (cq[j][k].x + cq[j][k].y) + (cq[j][k].z + cq[j][k].w);
      wsum[j] = LN ? wave_sum(t) : 0.f;
    }
  };
  sum_cls();
  char* small_base = ring + NSLOT * slot_bytes;      // LN: [NSLOT][NW][64 floats] of {mean, rstd} pairs

  // ---- producer: one ring item = one tile of TT token rows -------------------------------
  int pi = 0, pimg = 0, ptile = 0, pslot = 0;
  const char* xbytes = reinterpret_cast<const char*>(p.x);
  const char* psrc = xbytes + EP_IMG_OFF(p, wg) * ES;
  auto produce = [&]() {
    if (pi < n_items) {
      const int left = N - ptile * TT;
      const unsigned limit = (unsigned)((left < TT ? left : TT) * rowbytes - 16);
      dma_rows<NW, KDMA>(psrc, limit, ring + pslot * slot_bytes, npiece, w, lane16);
      if (LN) {
        const int bb = (wg + pimg * G) < p.B ? (wg + pimg * G) : wg;
        int e = ptile * TT * 2 + lane; e = e < 2 * N ? e : 2 * N - 1;
        const float* ts = p.tokstat + (int64_t)(p.index ? p.index[bb] : bb) * N * 2 + e;
        __builtin_amdgcn_global_load_lds((gptr_t)ts, (lds_ptr_t)(small_base + (pslot * NW + w) * 256), 4, 0, 0);
      }
      ++pi;
      pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
      if (++ptile == tiles_per_img) {
        ptile = 0; ++pimg;
        psrc = xbytes + EP_IMG_OFF(p, (wg + pimg * G) < p.B ? (wg + pimg * G) : wg) * ES;
      } else {
        psrc += slot_bytes;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 acc[QW][KP];
  float m[QW], mL[QW], lsum[QW], c2[QW];
  int cslot = 0, i = 0;
  // wait for item i, free the slot of item i-1, refill it.  Outstanding VMEM ops of this wave, oldest first: DMA of
  // items i..pi-1 (KD each), then the S / P stores of the previous iteration.  Requiring <= (pi-1-i)*KD outstanding
  // retires item i (and, harmlessly early, a few ops of item i+1 in place of the stores).
  auto ring_step = [&]() {
    const int ahead = pi - 1 - i;
    if (ahead == NSLOT - 2) wait_vmcnt_imm<(NSLOT - 2) * KD>();
    else wait_vmcnt(ahead * KD);
    ring_barrier();
    produce();
  };
  if (q0 >= Q || EP_STREAM_ABLATE == 1) {
    // a wave without a query (Q < QW * NW) only keeps the ring turning
    for (; i < n_items; ++i) ring_step();
    return;
  }
  // Loop shape (it decides what the register allocator does with the 24 accumulator registers): image-outer /
  // tile-inner, so the accumulators are initialised in front of the tile loop and stored behind it -- plain
  // loop-carried values, no conditional re-initialisation inside the loop; and the rest of a mini-batch is written out
  // on BOTH sides of the rare max-rescale branch, so each side updates the accumulators in place.  The flat item loop
  // with `if (first tile) acc = 0`, `if (rescale) acc *= f` and `if (last tile) store` made hipcc copy all accumulators
  // twice per tile (27 v_mov_b64 of ~300 instructions).
  for (int cimg = 0; cimg < n_img; ++cimg) {
    const int b = wg + cimg * G;
    if (p.cls_bstride != 0 && cimg != 0) { load_cls(b); sum_cls(); }   // per-image query override (rare path)
    float* Srow[QW];                                 // where this lane's row of a mini-batch stores its raw scores
#pragma unroll
    for (int j = 0; j < QW; ++j) Srow[j] = p.S + ((int64_t)b * Q + (q0 + j < Q ? q0 + j : q0)) * N + (lane >> 4);
#pragma unroll
    for (int j = 0; j < QW; ++j) {
      m[j] = -INFINITY; mL[j] = -INFINITY; lsum[j] = 0.f; c2[j] = 0.f;
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[j][k] = f4{0.f, 0.f, 0.f, 0.f};
    }
    for (int ctile = 0; ctile < tiles_per_img; ++ctile, ++i) {
      ring_step();
      const int n0 = ctile * TT;
      const int nvalid = (N - n0) < TT ? (N - n0) : TT;
      const char* tile = ring + cslot * slot_bytes;
      const float* small = reinterpret_cast<const float*>(small_base + (cslot * NW + w) * 256);
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
      // ---- compute: TT/TB butterfly mini-batches per tile ------------------------------------
#pragma unroll
      for (int t0 = 0; t0 < TT; t0 += TB) {
        if (t0 >= nvalid) break;
        f4 xv[TB][KP];
        load_rows<QW, KP, BF16>(tile + t0 * rowbytes, rowbytes, coff, xv, BF16 && p.x_bf16 == 2);     // (x_bf16 == 2: fp16-stored tokens)
        float part[QW][TB];
        partial_scores<QW, KP>(cq, xv, part);
        float u[QW];
        butterfly<QW>(part, u);
        const int row = lane >> 4;
        const bool rowvalid = (t0 + row) < nvalid;
        float tmean = 0.f, trstd = 1.f;
        if (LN) {
          tmean = small[2 * (t0 + row)]; trstd = small[2 * (t0 + row) + 1];
#pragma unroll
          for (int j = 0; j < QW; ++j) u[j] = trstd * (u[j] - tmean * wsum[j]);     // q . xhat
        }
        float ue[QW];
        bool need = false;
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          ue[j] = rowvalid ? u[j] : -INFINITY;
          need |= ue[j] > m[j] + LAZY_MAX_THR;
        }
        // weights, raw scores out, pooling: the part of the mini-batch behind the rescale decision
        auto finish = [&]() {
          float pr[QW];
#pragma unroll
          for (int j = 0; j < QW; ++j) {
            pr[j] = __builtin_amdgcn_exp2f(fmaf(ue[j], LOG2E, -mL[j]));      // invalid rows: exp2(-inf) = 0
            lsum[j] += pr[j];
          }
          if ((lane & 15) == 0 && rowvalid) {                                // raw scores for backward / attention maps
#pragma unroll
            for (int j = 0; j < QW; ++j)
              if (q0 + j < Q) Srow[j][n0 + t0] = u[j];
          }
          if (LN) {
#pragma unroll
            for (int j = 0; j < QW; ++j) { pr[j] *= trstd; c2[j] = fmaf(pr[j], tmean, c2[j]); }   // weights a * rstd
          }
          if (EP_STREAM_ABLATE != 3) accumulate_rows<QW, KP>(pr, xv, acc);
        };
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {     // wave-uniform, rare
#pragma unroll
          for (int j = 0; j < QW; ++j) {
            const float mx = fmaxf(fmaxf(readlane_f(ue[j], 0), readlane_f(ue[j], 16)),
                                   fmaxf(readlane_f(ue[j], 32), readlane_f(ue[j], 48)));
            const float mn = fmaxf(m[j], mx);
            const float f = __builtin_amdgcn_exp2f((m[j] - mn) * LOG2E);   // m = -inf -> 0
            m[j] = mn; mL[j] = mn * LOG2E;
            lsum[j] *= f; c2[j] *= f;
            // in place, by hand: with `acc[j][k] *= f` here hipcc renames the accumulators across this rare branch and
            // copies them back at the loop latch of EVERY tile (10 vector moves) -- or, with the rest of the mini-batch
            // written once behind the branch, twice per tile.  The volatile asm pins them to their registers; the rest of
            // the mini-batch then follows once, and the kernel needs 128 instead of 162 registers.
            const f2 ff = {f, f};
#pragma unroll
            for (int k = 0; k < KP; ++k) {
              f2 lo = acc[j][k].xy, hi = acc[j][k].zw;
              asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(lo) : "v"(ff));
              asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(hi) : "v"(ff));
              acc[j][k].xy = lo; acc[j][k].zw = hi;
            }
          }
        }
        finish();
      }
    }
    // ---- image epilogue -------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < QW; ++j) {
      const float l = readlane_f(lsum[j], 0) + readlane_f(lsum[j], 16) +
                      readlane_f(lsum[j], 32) + readlane_f(lsum[j], 48);
      const float inv = 1.0f / l;
      const float shift = LN ? (readlane_f(c2[j], 0) + readlane_f(c2[j], 16) + readlane_f(c2[j], 32) +
                                readlane_f(c2[j], 48)) * inv : 0.f;
      if (q0 + j < Q) {
        float* Pq = p.P + ((int64_t)b * Q + q0 + j) * D;
        bool handed = false;
        if constexpr (IPOK) handed = p.ip_ycnt != nullptr;
        if (handed) {
          // these rows are read by other workgroups inside this launch (ep_inpass.h): write-through 16-byte stores
          if constexpr (IPOK) {
            const __amdgpu_buffer_rsrc_t rP = ip_rsrc(p.P, (size_t)p.B * Q * D * sizeof(float));
            const unsigned off = (unsigned)(((int64_t)b * Q + q0 + j) * D * sizeof(float));
#pragma unroll
            for (int k = 0; k < KP; ++k) {
              const int c = lane + 64 * k;
              const f4 v = acc[j][k] * inv - shift;
              if (c < nchunk) ip_store16_wt(rP, off + 16u * (unsigned)c, f4v{v.x, v.y, v.z, v.w});
            }
          }
        } else {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const int c = lane + 64 * k;
          if (c < nchunk) *reinterpret_cast<f4*>(Pq + 4 * c) = acc[j][k] * inv - shift;
        }
        }
        if (lane == 0) {
          const f4 rec = {m[j], l, 0.f, 0.f};
          *reinterpret_cast<f4*>(p.ML + ((int64_t)b * Q + q0 + j) * 4) = rec;
        }
      }
    }
    if constexpr (IPOK) {
      if (p.ip_ycnt) ip_arrive(p.ip_ycnt + (b >> 5) * IP_CNT_STRIDE);       // this wave's P rows of image b are out
    }
  }
  // ---- in-pass value projection (ep_inpass.h): this workgroup has streamed all its images; while the others still
  // stream it runs projection tasks on the idle matrix pipe.  Rounds: R = ceil(B / G); the `nfull` workgroups that own an
  // image of the last round are busy until the end of the pass, so the tasks of their EARLIER images are dealt to the
  // workgroups that are done one round sooner ("helpers"); a workgroup runs the tasks of the images it finished last.
  if constexpr (IPOK) {
    if (p.ip_ycnt) {
      __syncthreads();                                       // every wave has left the token ring
      stamp(1);
      const int R = (p.B + G - 1) / G;
      const int nfull = p.B - (R - 1) * G;                   // workgroups with R images (== G when B % G == 0)
      const int nh = G - nfull;                              // helpers: done one round before the end of the pass
      // Tasks.  Row blocks below ip_yr0 (= (R - 1) G when the BatchNorm kernel has that split, else 0) belong to images
      // that end a round before the pass does: their y is computed whole (all four K quarters in one task, 8 tasks per
      // row block) by the helpers, hidden under the last round.  The row blocks of the LAST round are on the critical
      // path: four K-quarter tasks per (row block, query), one per owner, summed by the BatchNorm kernel.
      const int r0 = p.ip_yr0;
      for (int b = wg; b < p.B; b += G) {                    // this workgroup's images of the last round (all of them if r0 == 0)
        if (b < r0) continue;
        ip_wait<false>(p.ip_ycnt + (b >> 5) * IP_CNT_STRIDE, IP_TARGET, p.ip_err);   // the task reads P with sc1 loads
        stamp(2);
        const int r = b & 31;
        ip_y_task<KP>(p, b >> 5, r >> 2, r & 3, 1, p.ip_ypart + (int64_t)(r & 3) * p.B * D, ring);
        stamp(3);
      }
      if (wg >= nfull && nh > 0)
        for (int t = wg - nfull; t < r0 / 4; t += nh) {      // whole-y task t: row block t / 8, query t % 8
          ip_wait<false>(p.ip_ycnt + (t >> 3) * IP_CNT_STRIDE, IP_TARGET, p.ip_err);
          stamp(2);
          ip_y_task<KP>(p, t >> 3, t & 7, 0, 4, p.ip_y, ring);
          stamp(3);
        }
      stamp(4);
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward (gradient of cls_token)
// ring items per image: H = ceil(Q/HR) header items holding HR fp32 rows of dP[b] each, then the token tiles.
// Every item additionally carries one 4-byte-per-lane DMA per wave into a private 256-byte area:
// header items fetch ML[b,q,0:4] of the wave's queries, token items fetch S[b,q,n0:n0+TT].
// ---------------------------------------------------------------------------------------
template <int QW, int KP, int NW, int DFIX, bool BF16, bool LN>
__global__ __launch_bounds__(NW * 64, (stream_waves_per_cu(QW, KP, NW) / 4)) void ep_pool_bwd_kernel(PoolParams p, SideTasks side) {
  using Cfg = StreamCfgT<QW, KP, NW>;
  constexpr int NSLOT = Cfg::NSLOT_B, KDMA = Cfg::KDMA, TT = Cfg::TT * (BF16 ? 2 : 1);
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int HR = Cfg::TT;                       // fp32 dP rows that fit one ring slot (header items)
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  // Block order = dispatch order: pooling workgroups [0, first_block), then the side tasks, then the remaining
  // pooling workgroups.  Default first_block = G: the side tasks fill the tail of the pass.  Starting them earlier
  // (EP_SIDE_EARLY=<first_block>, diagnostics) measured slower at every position (0.495-0.520 vs 0.489 ms per step at
  // 256x768): dispatch is in order, so pooling workgroups queued behind the side tasks start late.
  const int G = gridDim.x - side.total;             // pooling workgroups
  int wg = blockIdx.x;
  if constexpr (NW == 4) {
    if (wg >= side.first_block) {
      if (wg < side.first_block + side.total) {
        // dy, an operand of the dWv side tasks, may be written inside this launch (folded BatchNorm backward of the dP
        // tasks): wait until every row block's tasks have arrived (they run in front of the token stream, normally ~100 us
        // before a side task gets a CU slot -- but not with a small pooling grid), then one acquire per workgroup
        if (p.ip_fold_dz) {
          const int nrb = p.B >> 5;
          if (threadIdx.x < 64) {
            int spins = 0;
            for (;;) {
              bool ok = true;
              for (int r = threadIdx.x; r < nrb; r += 64)
                ok &= __hip_atomic_load(p.ip_dcnt + r * IP_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= IP_TARGET;
              if (__all(ok)) break;
              if (++spins > IP_SPIN_LIMIT) { if (threadIdx.x == 0 && p.ip_err) atomicAdd(p.ip_err, 1); break; }
              __builtin_amdgcn_s_sleep(64);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
        }
        run_side_task(side, wg - side.first_block, ring);
        return;
      }
      wg -= side.total;
    }
  }
  const int lane = lane_id();
  const int w = wave_id_uniform();
  // ---- in-pass contractions (ep_inpass.h): clear the first pass's counters; produce the dP rows of this workgroup's
  // images' tasks, publish them, and wait for the row blocks of its own images before the stream (whose header items
  // read dP) starts.  The producers of a row block sit in the consumer's own aligned group of 32 workgroups.
  constexpr bool IPOK = (NW == 4 && QW == 2 && KP <= 3 && !LN);
  if constexpr (IPOK) {
    if (wg == 0 && p.ip_zero)
      for (int t = threadIdx.x; t < p.ip_nzero; t += NW * 64) p.ip_zero[t] = 0;
    auto stamp = [&](int k) {                              // diagnostic only (EP_IP_STAMP)
      if (p.dbg && threadIdx.x == 0) p.dbg[(int64_t)wg * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    if (p.ip_dy) {
      stamp(0);
      // Tasks: image b names task b.  A workgroup runs the task of its FIRST image itself; the tasks of later-round
      // images go to the workgroups that stream the fewest images (the last ones of the grid, which the SIMDs serve last
      // anyway: wave arbitration is oldest-first), so the workgroups with the most images start streaming first.
      const int R = (p.B + G - 1) / G;
      const int nfull = p.B - (R - 1) * G;                   // workgroups with R images
      const int nh = G - nfull;                              // helpers
      int first = 1;
      auto run = [&](int b) {
        ip_dp_task<KP>(p, b, ring);
        if (first) { stamp(1); first = 0; }
        ip_arrive(p.ip_dcnt + (b >> 5) * IP_CNT_STRIDE);
      };
      run(wg);
      if (nh == 0) { for (int b = wg + G; b < p.B; b += G) run(b); }
      else if (wg >= nfull)                                  // later-round image G + e goes to helper G - 1 - (e % nh)
        for (int e = G - 1 - wg; e < p.B - G; e += nh) run(G + e);
      stamp(2);
      for (int b = wg; b < p.B; b += G) ip_wait<false>(p.ip_dcnt + (b >> 5) * IP_CNT_STRIDE, IP_TARGET, p.ip_err);
      stamp(3);
    }
  }
  const int D = DFIX ? DFIX : p.D;
  const int N = p.N, Q = p.Q;
  const int rowbytes = D * ES;                      // token rows
  const int hrowbytes = D * 4;                      // dP rows (always fp32)
  const int nchunk = D >> 2;
  const int slot_bytes = TT * rowbytes;
  const int npiece = slot_bytes >> 10;
  const int tiles_per_img = (N + TT - 1) / TT;
  const int H = (Q + HR - 1) / HR;
  const int HD = p.dyv ? 1 : 0;                     // one more header item: the rows dy[b], y[b] the delta terms come from
  const int H2 = H + HD;
  const int items_per_img = H2 + tiles_per_img;
  const int dv_row = p.Dv * 4;                      // bytes of a dy / y row
  const int dv_rp = (dv_row + 1023) >> 10;          // 1-KiB ring pieces per row: dy at piece 0, y at piece dv_rp
  const int dvq = HD ? p.Dv / Q : 0;                // value channels per query
  const int n_img = (p.B - wg + G - 1) / G;
  const int n_items = n_img * items_per_img;
  const int q0 = w * QW;
  const unsigned lane16 = (unsigned)lane * 16u;
  char* small_base = ring + NSLOT * slot_bytes;          // [NSLOT][NW][64 floats]
  int coff[KP], hoff[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int c = (lane + 64 * k) < nchunk ? (lane + 64 * k) : 0;
    coff[k] = c * (4 * ES); hoff[k] = c * 16;
  }
  const char* xbytes = reinterpret_cast<const char*>(p.x);

  f4 gacc[QW][KP];
  float c3[QW];                      // LN: sum dS rstd mean (the mean term of sum dS xhat)
#pragma unroll
  for (int j = 0; j < QW; ++j) {
    c3[j] = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) gacc[j][k] = f4{0.f, 0.f, 0.f, 0.f};
  }
  static_assert(!LN || QW * TT + 2 * TT <= 64, "LN: scores and token statistics share one 64-lane small piece");

  if (n_items > 0) {
    // lane -> element of the per-item small DMA
    int hq = q0 + (lane >> 2); hq = hq < Q ? hq : Q - 1;           // header: ML[b, hq, lane & 3]
    int sq = q0 + lane / TT; sq = sq < Q ? sq : Q - 1;             // tile:   S[b, sq, n0 + lane % TT]
    const int st = lane % TT;
    int pi = 0, pimg = 0, pidx = 0, pslot = 0;
    // per-image bases of the producer (recomputed when it moves to the next image, not per ring item)
    const char* pimg_x = nullptr; const char* pimg_dP = nullptr; const float* pimg_ML = nullptr;
    const float* pimg_S = nullptr; const float* pimg_ts = nullptr;
    const char* pimg_dy = nullptr; const char* pimg_y = nullptr;
    auto producer_image = [&]() {
      const int b = (wg + pimg * G) < p.B ? (wg + pimg * G) : wg;
      pimg_x = xbytes + EP_IMG_OFF(p, b) * ES;
      pimg_dP = reinterpret_cast<const char*>(p.dP + (int64_t)b * Q * D);
      pimg_ML = p.ML + ((int64_t)b * Q + hq) * 4 + (lane & 3);
      pimg_S = p.S + ((int64_t)b * Q + sq) * N;
      if (LN) pimg_ts = p.tokstat + (int64_t)(p.index ? p.index[b] : b) * N * 2;
      if (HD) {
        pimg_dy = reinterpret_cast<const char*>(p.dyv + (int64_t)b * p.Dv);
        pimg_y = reinterpret_cast<const char*>(p.yv + (int64_t)b * p.Dv);
      }
    };
    producer_image();
    auto produce = [&]() {
      if (pi < n_items) {
        char* small = small_base + (pslot * NW + w) * 256;
        char* slot = ring + pslot * slot_bytes;
        if (pidx < H) {                                   // header: rows of dP[b]
          const int r0 = pidx * HR;
          const int rows = (Q - r0) < HR ? (Q - r0) : HR;
          // (rows written inside this launch by the in-pass dP tasks are read with sc1: no acquire needed, ep_inpass.h)
          if (IPOK && p.ip_dy) dma_rows<NW, KDMA, IP_SC1>(pimg_dP + (int64_t)r0 * hrowbytes, (unsigned)(rows * hrowbytes - 16), slot, npiece, w, lane16);
          else dma_rows<NW, KDMA>(pimg_dP + (int64_t)r0 * hrowbytes, (unsigned)(rows * hrowbytes - 16), slot, npiece, w, lane16);
          __builtin_amdgcn_global_load_lds((gptr_t)pimg_ML, (lds_ptr_t)small, 4, 0, 0);
        } else if (pidx < H2) {                           // delta item: dy[b] | y[b] (same instruction count as every item)
#pragma unroll
          for (int j = 0; j < KDMA; ++j) {
            int pc = w + NW * j;
            pc = pc < 2 * dv_rp ? pc : 2 * dv_rp - 1;
            const bool second = pc >= dv_rp;
            unsigned off = (unsigned)(second ? pc - dv_rp : pc) * 1024u + lane16;
            off = off < (unsigned)(dv_row - 16) ? off : (unsigned)(dv_row - 16);
            // (dy written inside this launch by the folded BatchNorm backward of the dP tasks is read with sc1)
            if (IPOK && p.ip_fold_dz) __builtin_amdgcn_global_load_lds((gptr_t)((second ? pimg_y : pimg_dy) + off), (lds_ptr_t)(slot + pc * 1024), 16, 0, IP_SC1);
            else __builtin_amdgcn_global_load_lds((gptr_t)((second ? pimg_y : pimg_dy) + off), (lds_ptr_t)(slot + pc * 1024), 16, 0, 0);
          }
          __builtin_amdgcn_global_load_lds((gptr_t)pimg_ML, (lds_ptr_t)small, 4, 0, 0);
        } else {
          const int n0 = (pidx - H2) * TT;
          const int rows = (N - n0) < TT ? (N - n0) : TT;
          dma_rows<NW, KDMA>(pimg_x + (int64_t)n0 * rowbytes, (unsigned)(rows * rowbytes - 16), slot, npiece, w, lane16);
          int nn = n0 + st; nn = nn < N ? nn : N - 1;
          const float* ss = pimg_S + nn;
          if (LN && lane >= QW * TT) {              // lanes behind the scores fetch the {mean, rstd} pairs of the tile
            int e = n0 * 2 + (lane - QW * TT); e = e < 2 * N ? e : 2 * N - 1;
            ss = pimg_ts + e;
          }
          __builtin_amdgcn_global_load_lds((gptr_t)ss, (lds_ptr_t)small, 4, 0, 0);
        }
        ++pi;
        pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
        if (++pidx == items_per_img) { pidx = 0; ++pimg; producer_image(); }
      }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) produce();

    f4 gq[QW][KP];                  // dP rows of this wave's queries for the current image
    float mLq[QW], il[QW], dl[QW];  // row max * log2e, 1/l, delta of this wave's queries
    float gsum[QW];                 // LN: sum over D of the dP row
    int cslot = 0, i = 0;
    constexpr int KD = KDMA + 1;
    const char* tile = nullptr;
    const float* small = nullptr;
    // wait for item i, free the slot of item i-1, refill it (see the forward pass)
    auto ring_step = [&]() {
      const int ahead = pi - 1 - i;
      if (ahead == NSLOT - 2) wait_vmcnt_imm<(NSLOT - 2) * KD>();
      else wait_vmcnt(ahead * KD);
      ring_barrier();
      produce();
      tile = ring + cslot * slot_bytes;
      small = reinterpret_cast<const float*>(small_base + (cslot * NW + w) * 256);
      cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    };
    if (q0 >= Q) {
      // a wave without a query (Q < QW * NW) only keeps the ring turning; its accumulators stay zero and are not stored
      for (; i < n_items; ++i) ring_step();
    } else
    // Loop shape as in the forward pass: image-outer, the header items, then the token tiles in a loop of their own whose
    // body updates the accumulators unconditionally and in place.  The flat item loop (header / delta / tile chosen per
    // item, tile work skipped for query-less waves) made hipcc copy all 24 accumulator registers in front of every tile.
    for (int cimg = 0; cimg < n_img; ++cimg) {
      for (int cidx = 0; cidx < H2; ++cidx, ++i) {
        ring_step();
        if (cidx < H) {
          // header item: pick up the dP rows of my queries that live in this item
#pragma unroll
          for (int j = 0; j < QW; ++j) {
            const int q = q0 + j;
            if (q < Q && q / HR == cidx) {
              const int r = q % HR;
#pragma unroll
              for (int k = 0; k < KP; ++k) {
                f4 v = *reinterpret_cast<const f4*>(tile + r * hrowbytes + hoff[k]);
                if (lane + 64 * k >= nchunk) v = f4{0.f, 0.f, 0.f, 0.f};
                gq[j][k] = v;
              }
            } else if (q >= Q && cidx == 0) {
#pragma unroll
              for (int k = 0; k < KP; ++k) gq[j][k] = f4{0.f, 0.f, 0.f, 0.f};
            }
            mLq[j] = small[4 * j + 0] * LOG2E;
            il[j] = 1.0f / small[4 * j + 1];
            dl[j] = small[4 * j + 2];
          }
          if (LN && cidx == H - 1) {
#pragma unroll
            for (int j = 0; j < QW; ++j) {
              float t = 0.f;
#pragma unroll
              for (int k = 0; k < KP; ++k) t += (gq[j][k].x + gq[j][k].y) + (gq[j][k].z + gq[j][k].w);
              gsum[j] = wave_sum(t);
            }
          }
        } else {
          // delta item: dl = dy[b, q-slice] . y[b, q-slice] for this wave's queries (overrides the ML[.,.,2] picked up above)
#pragma unroll
          for (int j = 0; j < QW; ++j) {
            const int q = q0 + j < Q ? q0 + j : Q - 1;
            float t = 0.f;
            for (int c = 4 * lane; c < dvq; c += 256) {
              const f4 a = *reinterpret_cast<const f4*>(tile + (q * dvq + c) * 4);
              const f4 b = *reinterpret_cast<const f4*>(tile + dv_rp * 1024 + (q * dvq + c) * 4);
              t += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
            }
            dl[j] = wave_sum(t);
          }
        }
      }
      for (int ctile = 0; ctile < tiles_per_img; ++ctile, ++i) {
        ring_step();
        const int n0 = ctile * TT;
        const int nvalid = (N - n0) < TT ? (N - n0) : TT;      // >= 1
#pragma unroll
        for (int t0 = 0; t0 < TT; t0 += TB) {
          if (t0 > 0 && t0 >= nvalid) break;                    // (TT > TB only: later mini-batches of a ragged tile)
          f4 xv[TB][KP];
          load_rows<QW, KP, BF16>(tile + t0 * rowbytes, rowbytes, coff, xv);
          float part[QW][TB];
          partial_scores<QW, KP>(gq, xv, part);
          float u[QW];
          butterfly<QW>(part, u);                       // dA[q][t] in row t
          const int row = lane >> 4;
          const bool rowvalid = (t0 + row) < nvalid;
          float tmean = 0.f, trstd = 1.f;
          if (LN) {
            tmean = small[QW * TT + 2 * (t0 + row)]; trstd = small[QW * TT + 2 * (t0 + row) + 1];
#pragma unroll
            for (int j = 0; j < QW; ++j) u[j] = trstd * (u[j] - tmean * gsum[j]);     // dA = dP . xhat
          }
          float wgt[QW];
#pragma unroll
          for (int j = 0; j < QW; ++j) {
            const float s = small[j * TT + t0 + row];
            const float a = __builtin_amdgcn_exp2f(fmaf(s, LOG2E, -mLq[j])) * il[j];
            wgt[j] = rowvalid ? a * (u[j] - dl[j]) : 0.f;
            if (LN) { wgt[j] *= trstd; c3[j] = fmaf(wgt[j], tmean, c3[j]); }
          }
          accumulate_rows<QW, KP>(wgt, xv, gacc);
        }
      }
    }
  }
  if constexpr (IPOK) { if (p.dbg && p.ip_dy && threadIdx.x == 0) p.dbg[(int64_t)wg * 8 + 4] = __builtin_amdgcn_s_memrealtime(); }
  // per-workgroup partial of sum_b sum_n dS x  (reduced + scaled by ep_reduce_partials)
#pragma unroll
  for (int j = 0; j < QW; ++j)
    if (q0 + j < Q) {
      float* Gq = p.Gpart + ((int64_t)wg * Q + q0 + j) * D;
      const float shift = LN ? (readlane_f(c3[j], 0) + readlane_f(c3[j], 16)) + (readlane_f(c3[j], 32) + readlane_f(c3[j], 48)) : 0.f;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int c = lane + 64 * k;
        if (c < nchunk) *reinterpret_cast<f4*>(Gq + 4 * c) = gacc[j][k] - shift;
      }
    }
}

// ---------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------
// stream_resident_blocks_per_cu(): launch_one answers with the occupancy of the kernel it WOULD launch instead of launching
static thread_local int* t_occ_query = nullptr;

template <int QW, int KP, int NW, int DFIX, bool BF16, bool LN>
static int launch_one(bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side) {
  using Cfg = StreamCfgT<QW, KP, NW>;
  const int D = p.D;
  const size_t slot = (size_t)Cfg::TT * D * 4;          // bf16: twice the tokens, half the bytes each
  size_t lds = (bwd || LN) ? (size_t)Cfg::NSLOT_B * (slot + (size_t)NW * 256) : (size_t)Cfg::NSLOT_F * slot;
  if constexpr (NW == 4 && QW == 2 && KP <= 3 && !LN) {     // in-pass tasks reuse the ring's LDS
    if (bwd && p.ip_dy && lds < ip_dp_lds_bytes(KP)) lds = ip_dp_lds_bytes(KP);
    if (!bwd && p.ip_ycnt && lds < ip_y_lds_bytes(KP)) lds = ip_y_lds_bytes(KP);
  }
  SideTasks sd{};
  if (bwd && side && side->total > 0) {
    if (NW != 4) { set_error("side tasks need 4-wave workgroups"); return EP_E_UNSUPPORTED; }
    sd = *side;
    // The bf16 x3 tile (ep_wgrad3.h) halves the matrix time of a side contraction but adds ~110 vector instructions per
    // K-tile (the split) -- and these passes are vector-issue co-limited.  Measured at 256 x 768 (544 tiles, ~30 us of side
    // work overlapping the drain of the pooling workgroups): second pass 195 against 191 us with it, the step 0.424 against
    // 0.422 ms; with the ~1600 tiles of the SigLIP head (most of them run after the stream has ended) 0.832 against 0.858 ms.
    // So: only when the side work is long against the drain.  EP_SIDE_B3=0 / 1 forces.
    static int b3_env = -2;
    if (b3_env == -2) { const char* e = getenv("EP_SIDE_B3"); b3_env = e ? atoi(e) : -1; }
    if (b3_env >= 0) sd.b3 = sd.b3 && b3_env;
    else if (sd.total <= 1024 && sd.g[0].nterms != 1) sd.b3 = 0;    // (AMP-bf16: the single-product tile has no split to pay for)
    if (lds < SIDE_LDS_BYTES) lds = SIDE_LDS_BYTES;
    static int early = -1;
    if (early < 0) { const char* e = getenv("EP_SIDE_EARLY"); early = e ? atoi(e) : 0; }
    sd.first_block = (early > 0 && early < grid) ? early : grid;
  } else {
    sd.first_block = grid;
  }
  auto kf = ep_pool_fwd_kernel<QW, KP, NW, DFIX, BF16, LN>;
  auto kb = ep_pool_bwd_kernel<QW, KP, NW, DFIX, BF16, LN>;
  const void* fn = bwd ? (const void*)kb : (const void*)kf;
  // diagnostic only (EP_IP_STAMP=1): per-workgroup timeline of the in-pass tasks, printed to stderr (synchronises!)
  static int stampmode = -1;
  if (stampmode < 0) { const char* e = getenv("EP_IP_STAMP"); stampmode = e ? atoi(e) : 0; }
  const bool stamped = stampmode && ((bwd && p.ip_dy) || (!bwd && p.ip_ycnt)) && grid <= 4096;
  if (stamped) {
    static unsigned long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 4096 * 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg, 0, (size_t)grid * 8 * sizeof(unsigned long long), st);
    PoolParams q = p;
    q.dbg = dbg;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (bwd) hipLaunchKernelGGL(kb, dim3(grid + sd.total), dim3(NW * 64), lds, st, q, sd);
    else hipLaunchKernelGGL(kf, dim3(grid), dim3(NW * 64), lds, st, q);
    (void)hipStreamSynchronize(st);
    static unsigned long long host[4096 * 8];
    (void)hipMemcpy(host, dbg, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    static int printed[2] = {0, 0};
    if (printed[bwd ? 1 : 0]++ % 20 == 10) {
      unsigned long long t0 = ~0ull;
      for (int g = 0; g < grid; ++g) if (host[g * 8] && host[g * 8] < t0) t0 = host[g * 8];
      fprintf(stderr, "[EP_IP_STAMP] %s grid %d: mean us since the first workgroup started, per third of the grid\n", bwd ? "bwd" : "fwd", grid);
      for (int part = 0; part < 3; ++part) {
        const int g0 = grid * part / 3, g1 = grid * (part + 1) / 3;
        double m[6] = {0}, mx[6] = {0}; int cnt[6] = {0};
        for (int g = g0; g < g1; ++g)
          for (int k = 0; k < 5; ++k)
            if (host[g * 8 + k]) { const double us = (double)(host[g * 8 + k] - t0) / 100.0; m[k] += us; if (us > mx[k]) mx[k] = us; ++cnt[k]; }
        fprintf(stderr, "   wg %4d..%4d:", g0, g1 - 1);
        for (int k = 0; k < 5; ++k) fprintf(stderr, "  s%d %7.1f (max %7.1f)", k, cnt[k] ? m[k] / cnt[k] : -1.0, mx[k]);
        fprintf(stderr, "\n");
      }
    }
    return 0;
  }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
  if (t_occ_query) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, NW * 64, lds) != hipSuccess) { (void)hipGetLastError(); nb = -1; }
    *t_occ_query = nb;
    return 0;
  }
  if (bwd) hipLaunchKernelGGL(kb, dim3(grid + sd.total), dim3(NW * 64), lds, st, p, sd);
  else hipLaunchKernelGGL(kf, dim3(grid), dim3(NW * 64), lds, st, p);
  EP_LAUNCH_CHECK(bwd ? "ep_pool_bwd_kernel" : "ep_pool_fwd_kernel");
  return 0;
}

template <int QW, int KP, int NW>
static int launch_cfg(bool bwd, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side) {
  using Cfg = StreamCfgT<QW, KP, NW>;
  if constexpr (!Cfg::VALID) {
    set_error("no streaming kernel for qw=%d kp=%d nw=%d", QW, KP, NW);
    return EP_E_UNSUPPORTED;
  } else {
    // compile-time D for the shapes the benchmark configs use (immediate LDS offsets)
    if constexpr ((QW == 2 && NW == 4 && (KP == 3 || KP == 4)) || (QW == 4 && NW == 8 && KP == 3)) {
      if (p.D == 256 * KP && !p.x_bf16 && !p.tokstat) return launch_one<QW, KP, NW, 256 * KP, false, false>(bwd, p, grid, st, side);
      if (p.D == 256 * KP && !p.tokstat) return launch_one<QW, KP, NW, 256 * KP, true, false>(bwd, p, grid, st, side);
    }
    if (p.tokstat) {
      if (p.x_bf16) {
        // bf16 tiles hold twice the tokens: scores and token statistics of a tile must still fit one 64-lane piece
        if constexpr (QW * 2 * Cfg::TT + 2 * 2 * Cfg::TT <= 64) return launch_one<QW, KP, NW, 0, true, true>(bwd, p, grid, st, side);
        else { set_error("LayerNorm-of-tokens mode on bf16 tokens: no kernel for qw=%d nw=%d", QW, NW); return EP_E_UNSUPPORTED; }
      }
      return launch_one<QW, KP, NW, 0, false, true>(bwd, p, grid, st, side);
    }
    if (p.x_bf16) return launch_one<QW, KP, NW, 0, true, false>(bwd, p, grid, st, side);
    return launch_one<QW, KP, NW, 0, false, false>(bwd, p, grid, st, side);
  }
}

template <int QW, int NW>
static int dispatch_kp(bool bwd, int kp, const PoolParams& p, int grid, hipStream_t st, const SideTasks* side) {
  switch (kp) {
    case 1: return launch_cfg<QW, 1, NW>(bwd, p, grid, st, side);
    case 2: return launch_cfg<QW, 2, NW>(bwd, p, grid, st, side);
    case 3: return launch_cfg<QW, 3, NW>(bwd, p, grid, st, side);
    case 4: return launch_cfg<QW, 4, NW>(bwd, p, grid, st, side);
    case 5: return launch_cfg<QW, 5, NW>(bwd, p, grid, st, side);
    case 6: return launch_cfg<QW, 6, NW>(bwd, p, grid, st, side);
  }
  set_error("no streaming kernel for kp=%d", kp);
  return EP_E_UNSUPPORTED;
}

bool stream_ln_supported(int D, int Q) { return stream_plan(1 << 20, 1, D, Q).ok; }
bool stream_ln_bf16_supported(int D, int Q) {
  const StreamPlan c = stream_plan(1 << 20, 1, D, Q);
  if (!c.ok) return false;
  const int tt = 2 * stream_tt(c.qw, c.kp, c.nw);
  return c.qw * tt + 2 * tt <= 64;
}

// Workgroups of the pass (with its in-pass tasks and side tasks, as `p` / `side` ask for them) that one CU holds at a time
// by registers and LDS; -1 when the runtime cannot tell.  The in-pass hand-off (ep_inpass.h) needs the whole pooling
// grid resident, so ep_pool.hip: pool_inpass_mask asks before it lets a workgroup wait on another one.
int stream_resident_blocks_per_cu(bool bwd, const StreamPlan& c, const PoolParams& p, const SideTasks* side) {
  int nb = -1;
  t_occ_query = &nb;
  const int rc = stream_launch(bwd, c, p, nullptr, side);
  t_occ_query = nullptr;
  return rc == 0 ? nb : -1;
}

int stream_launch(bool bwd, const StreamPlan& c, const PoolParams& p, hipStream_t st, const SideTasks* side) {
  if (c.qw == 1 && c.nw == 4) return dispatch_kp<1, 4>(bwd, c.kp, p, c.grid, st, side);
  if (c.qw == 1 && c.nw == 8) return dispatch_kp<1, 8>(bwd, c.kp, p, c.grid, st, side);
  if (c.qw == 2 && c.nw == 4) return dispatch_kp<2, 4>(bwd, c.kp, p, c.grid, st, side);
  if (c.qw == 2 && c.nw == 8) return dispatch_kp<2, 8>(bwd, c.kp, p, c.grid, st, side);
  if (c.qw == 4 && c.nw == 8) return dispatch_kp<4, 8>(bwd, c.kp, p, c.grid, st, side);
  set_error("no streaming kernel for qw=%d nw=%d", c.qw, c.nw);
  return EP_E_UNSUPPORTED;
}

}  // namespace ep
