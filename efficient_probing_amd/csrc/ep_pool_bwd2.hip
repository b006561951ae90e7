// EP attentive pooling, second token pass of the fused train step -- the TICKETED form (gfx950 / CDNA4).
//
//   dA = dP x^T ; dS = A (dA - delta) ; dcls = scale * sum_b dS x          (autograd of reference poolings/ep.py:35-44)
//
// Same arithmetic, ring and butterfly as ep_pool_bwd_kernel<2, KP, 4, ...> (ep_pool_stream.hip).  What differs is WHO
// streams WHICH image and what else happens inside the launch:
//
//   * images are handed out by TICKETS (one device-wide counter) instead of b = wg + j G: a workgroup that starts late or
//     runs slow simply takes fewer, so the launch can be laid out  [P1 pooling workgroups][weight-gradient side tasks]
//     [remaining pooling workgroups]  -- the side tasks (dWc, dWv, dbc: ep_side.h) get a CU slot from the first
//     microsecond and run on the idle matrix pipe UNDER the stream instead of extending the launch by 30-40 us behind it,
//     and the pooling workgroups queued behind them pick up whatever images are left when they arrive;
//   * the gradient partials are written PER IMAGE (Gpart[b]), so the result does not depend on which workgroup streamed
//     an image: bit-reproducible although the assignment is dynamic (the reduction sums images in fixed order);
//   * dP[b] = dy[b] Wv is produced inside the launch (ep_inpass.h: ip_dp_task, image b names task b of its 32-image row
//     block): a workgroup computes the task of its first image before it streams, and the task of its NEXT image at a
//     "task point" in the middle of the current one -- a tile index staggered over the workgroups, so that at any time only
//     a few of them are on the matrix pipe and the stream never stops chip-wide.  At the task point the ring is drained
//     (its LDS is the task's scratch), the running accumulators are parked in this image's partial slot and the dP rows
//     of the current image are re-read afterwards.  The row block of the next image is checked complete a few tiles before
//     the producer crosses into it (its 32 tickets were drawn within a few microseconds of each other, tens of
//     microseconds earlier).
//
// Shapes: Q = 8, D = 256 KP (KP = 1..3), projection width = D, B % 32 == 0, >= 20 token tiles per image, >= 64 pooling
// workgroups (host-checked: ep_pool.hip: bwd2_ok).
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_pool_stream.h"
#include "ep_stream_dev.h"
#include "ep_side.h"
#include "ep_inpass.h"
#include "ep_sidetask.h"

namespace ep {

template <int KP, int DFIX, bool BF16>
__global__ __launch_bounds__(256, 3) void ep_pool_bwd2_kernel(PoolParams p, SideTasks side) {
  constexpr int QW = 2, NW = 4;
  using Cfg = StreamCfgT<QW, KP, NW>;
  constexpr int NSLOT = Cfg::NSLOT_B, KDMA = Cfg::KDMA, TT = Cfg::TT * (BF16 ? 2 : 1), KD = KDMA + 1;
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int HR = Cfg::TT;                       // fp32 dP rows that fit one ring slot (header items)
  constexpr int Q = 8;
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  // block order = dispatch order: pooling workgroups [0, first_block), the side tasks, the remaining pooling workgroups
  const int G = gridDim.x - side.total;
  int wg = blockIdx.x;
  // diagnostic only (EP_IP_STAMP): 100 MHz timeline, 8 words per block
  auto stamp = [&](int blk, int k) { if (p.dbg && threadIdx.x == 0) p.dbg[(int64_t)blk * 8 + k] = __builtin_amdgcn_s_memrealtime(); };
  if (wg >= side.first_block) {
    if (wg < side.first_block + side.total) {
      stamp(G + wg - side.first_block, 0);
      run_side_task(side, wg - side.first_block, ring);
      stamp(G + wg - side.first_block, 1);
      return;
    }
    wg -= side.total;
  }
  stamp(wg, 0);
  unsigned long long t_task = 0;
  int n_imgs = 0;
  const int lane = lane_id();
  const int w = wave_id_uniform();
  if (wg == 0 && p.ip_zero)                         // this launch clears the first pass's counters
    for (int t = threadIdx.x; t < p.ip_nzero; t += NW * 64) p.ip_zero[t] = 0;
  const int D = DFIX ? DFIX : p.D;
  const int N = p.N, B = p.B;
  const int rowbytes = D * ES;                      // token rows
  const int hrowbytes = D * 4;                      // dP rows (always fp32)
  const int nchunk = D >> 2;
  const int slot_bytes = TT * rowbytes;
  const int npiece = slot_bytes >> 10;
  const int T = (N + TT - 1) / TT;                  // token tiles per image
  constexpr int H = (Q + HR - 1) / HR;              // header items holding the dP rows
  constexpr int H2 = H + 1;                         // + the delta item (dy[b] | y[b])
  const int items_per_img = H2 + T;
  const int dv_row = p.Dv * 4;
  const int dv_rp = (dv_row + 1023) >> 10;
  const int dvq = p.Dv / Q;
  const int q0 = w * QW;
  char* small_base = ring + NSLOT * slot_bytes;     // [NSLOT][NW][64 floats]
  int* tkw = reinterpret_cast<int*>(small_base + NSLOT * NW * 256);   // one broadcast word behind the ring
  int coff[KP];                                     // (D = 256 KP: every lane has a chunk of every 1-KiB piece)
#pragma unroll
  for (int k = 0; k < KP; ++k) coff[k] = (lane + 64 * k) * (4 * ES);
  const char* xbytes = reinterpret_cast<const char*>(p.x);
  const int tp = T / 4 + (int)((((unsigned)wg * 2654435761u) >> 10) % (unsigned)(T / 2));   // task tile, in [T/4, 3T/4)

  // one ticket = the index of an image nobody streams yet (or >= B: none left); uniform over the workgroup
  auto take_ticket = [&]() -> int {
    if (threadIdx.x == 0) tkw[0] = p.tick_base + __hip_atomic_fetch_add(p.tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = tkw[0];
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(t);
  };
  auto cnt_of = [&](int b) { return p.ip_dcnt + (b >> 5) * IP_CNT_STRIDE; };

  int b_cur = wg < p.tick_base ? wg : take_ticket();
  if (b_cur >= B) return;
  ip_dp_task_call<KP>(p.ip_dy, p.ip_Wv, const_cast<float*>(p.dP), B, b_cur, ring);
  ip_arrive(cnt_of(b_cur));
  ip_wait<false>(cnt_of(b_cur), IP_TARGET, p.ip_err);       // (the header DMA and the reloads below read dP with sc1)
  stamp(wg, 1);
  int b_next = B;                                            // known from the task point of the current image on

  // ---- producer: items (image pb, index pidx); it stops in front of the pending task point of its image ----
  int pi = 0, pb = b_cur, pidx = 0, pslot = 0, pstop = H2 + tp;
  bool pdone = false;
  // per-image bases of the producer: all wave-uniform (SGPRs); everything lane-dependent is recomputed per copy from the
  // lane index (see produce())
  const char* pimg_x = nullptr; const char* pimg_dP = nullptr; const float* pimg_ML = nullptr;
  const float* pimg_S = nullptr; const char* pimg_dy = nullptr; const char* pimg_y = nullptr;
  auto producer_image = [&]() {
    pimg_x = xbytes + EP_IMG_OFF(p, pb) * ES;
    pimg_dP = reinterpret_cast<const char*>(p.dP + (int64_t)pb * Q * D);
    pimg_ML = p.ML + (int64_t)pb * Q * 4;
    pimg_S = p.S + (int64_t)pb * Q * N;
    pimg_dy = reinterpret_cast<const char*>(p.dyv + (int64_t)pb * p.Dv);
    pimg_y = reinterpret_cast<const char*>(p.yv + (int64_t)pb * p.Dv);
  };
  producer_image();
  auto produce = [&]() {
    if (pdone || pidx == pstop) return;
    char* small = small_base + (pslot * NW + w) * 256;
    char* slot = ring + pslot * slot_bytes;
    // the per-piece lane offsets below are two vector instructions each; left to itself hipcc keeps one precomputed
    // offset (or 64-bit address) per DMA instruction of every item type alive across the token loop -- ~20 registers that
    // it then spills, reloading them (with s_waitcnt vmcnt(0)!) in front of every copy.  An opaque lane16 pins the
    // computation here.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const unsigned lane16 = (unsigned)ln * 16u;
    // lane -> element of the per-item small DMA: header ML[b, hq, lane & 3]; tile S[b, sq, n0 + lane % TT]
    int hq = q0 + (ln >> 2); hq = hq < Q ? hq : Q - 1;
    int sq = q0 + ln / TT; sq = sq < Q ? sq : Q - 1;
    const int st = ln % TT;
    const int ml_off = hq * 4 + (ln & 3);
    if (pidx < H) {                                   // header: rows of dP[b], written inside this launch -> sc1
      const int r0 = pidx * HR;
      const int rows = (Q - r0) < HR ? (Q - r0) : HR;
      dma_rows<NW, KDMA, IP_SC1>(pimg_dP + (int64_t)r0 * hrowbytes, (unsigned)(rows * hrowbytes - 16), slot, npiece, w, lane16);
      __builtin_amdgcn_global_load_lds((gptr_t)(pimg_ML + ml_off), (lds_ptr_t)small, 4, 0, 0);
    } else if (pidx < H2) {                           // delta item: dy[b] | y[b]
#pragma unroll
      for (int j = 0; j < KDMA; ++j) {
        int pc = w + NW * j;
        pc = pc < 2 * dv_rp ? pc : 2 * dv_rp - 1;
        const bool second = pc >= dv_rp;
        unsigned off = (unsigned)(second ? pc - dv_rp : pc) * 1024u + lane16;
        off = off < (unsigned)(dv_row - 16) ? off : (unsigned)(dv_row - 16);
        __builtin_amdgcn_global_load_lds((gptr_t)((second ? pimg_y : pimg_dy) + off), (lds_ptr_t)(slot + pc * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)(pimg_ML + ml_off), (lds_ptr_t)small, 4, 0, 0);
    } else {
      const int n0 = (pidx - H2) * TT;
      const int rows = (N - n0) < TT ? (N - n0) : TT;
      dma_rows<NW, KDMA>(pimg_x + (int64_t)n0 * rowbytes, (unsigned)(rows * rowbytes - 16), slot, npiece, w, lane16);
      int nn = n0 + st; nn = nn < N ? nn : N - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(pimg_S + (sq * N + nn)), (lds_ptr_t)small, 4, 0, 0);
    }
    ++pi;
    pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
    if (++pidx == items_per_img) {                    // (the task point of pb is behind us: b_next is known)
      if (b_next >= B) pdone = true;
      else { pb = b_next; pidx = 0; pstop = H2 + tp; producer_image(); }
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) produce();

  f4 gacc[QW][KP], gq[QW][KP];
  float mLq[QW], il[QW], dl[QW];
  int cslot = 0, ci = 0;
  const char* tile = nullptr;
  const float* small = nullptr;
  auto ring_step = [&]() {                            // wait for item ci, free the slot of item ci-1, refill it
    const int ahead = pi - 1 - ci;
    if (ahead == NSLOT - 2) wait_vmcnt_imm<(NSLOT - 2) * KD>();
    else wait_vmcnt(ahead * KD);
    ring_barrier();
    produce();
    tile = ring + cslot * slot_bytes;
    small = reinterpret_cast<const float*>(small_base + (cslot * NW + w) * 256);
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    ++ci;
  };
  const __amdgpu_buffer_rsrc_t rdP = ip_rsrc(p.dP, (size_t)B * Q * D * sizeof(float));

  for (;;) {                                          // one trip per image of this workgroup
    for (int cidx = 0; cidx < H2; ++cidx) {
      ring_step();
      if (cidx < H) {
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          const int q = q0 + j;
          if (q / HR == cidx) {
            const int r = q % HR;
#pragma unroll
            for (int k = 0; k < KP; ++k) gq[j][k] = *reinterpret_cast<const f4*>(tile + r * hrowbytes + 16 * (lane + 64 * k));
          }
          mLq[j] = small[4 * j + 0] * LOG2E;
          il[j] = 1.0f / small[4 * j + 1];
        }
      } else {
        // delta item: dl = dy[b, q-slice] . y[b, q-slice] for this wave's queries
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          const int q = q0 + j;
          float t = 0.f;
          for (int c = 4 * lane; c < dvq; c += 256) {
            const f4 a = *reinterpret_cast<const f4*>(tile + (q * dvq + c) * 4);
            const f4 bb = *reinterpret_cast<const f4*>(tile + dv_rp * 1024 + (q * dvq + c) * 4);
            t += (a.x * bb.x + a.y * bb.y) + (a.z * bb.z + a.w * bb.w);
          }
          dl[j] = wave_sum(t);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < QW; ++j)
#pragma unroll
      for (int k = 0; k < KP; ++k) gacc[j][k] = f4{0.f, 0.f, 0.f, 0.f};
    for (int ctile = 0; ctile < T; ++ctile) {
      if (ctile == tp) {
        // ---- task point: the producer stopped in front of this tile, the ring is empty.  Park the accumulators in this
        // image's partial slot, draw the next image and compute its dP rows, then pick everything up again.
        const unsigned long long tt0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          float* Gq = p.Gpart + ((int64_t)b_cur * Q + q0 + j) * D;
#pragma unroll
          for (int k = 0; k < KP; ++k) *reinterpret_cast<f4*>(Gq + 4 * (lane + 64 * k)) = gacc[j][k];
        }
        b_next = take_ticket();
        if (b_next < B) {
          ip_dp_task_call<KP>(p.ip_dy, p.ip_Wv, const_cast<float*>(p.dP), B, b_next, ring);
          ip_arrive(cnt_of(b_next));
        }
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          const float* Gq = p.Gpart + ((int64_t)b_cur * Q + q0 + j) * D;
          const unsigned off = (unsigned)(((int64_t)b_cur * Q + q0 + j) * D * sizeof(float));
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            gacc[j][k] = *reinterpret_cast<const f4*>(Gq + 4 * (lane + 64 * k));        // this lane's own stores
            const f4v v = ip_load16_coherent(rdP, off + 16u * (unsigned)(lane + 64 * k));   // handed-off rows: sc1
            gq[j][k] = f4{v.x, v.y, v.z, v.w};
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of ours is in flight before the ring restarts
#pragma unroll
        for (int j = 0; j < QW; ++j)
#pragma unroll
          for (int k = 0; k < KP; ++k) { asm volatile("" : "+v"(gacc[j][k])); asm volatile("" : "+v"(gq[j][k])); }
        pstop = items_per_img;
#pragma unroll
        for (int s = 0; s < NSLOT - 1; ++s) produce();
        if (p.dbg) t_task += __builtin_amdgcn_s_memrealtime() - tt0;
      }
      // the producer crosses into the next image three tiles from now: its dP row block must be complete by then
      if (ctile == T - NSLOT - 1 && b_next < B) ip_wait<false>(cnt_of(b_next), IP_TARGET, p.ip_err);
      ring_step();
      const int n0 = ctile * TT;
      const int nvalid = (N - n0) < TT ? (N - n0) : TT;
#pragma unroll
      for (int t0 = 0; t0 < TT; t0 += TB) {
        if (t0 > 0 && t0 >= nvalid) break;
        f4 xv[TB][KP];
        load_rows<QW, KP, BF16>(tile + t0 * rowbytes, rowbytes, coff, xv);
        float part[QW][TB];
        partial_scores<QW, KP>(gq, xv, part);
        float u[QW];
        butterfly<QW>(part, u);                       // dA[q][t] in row t
        const int row = lane >> 4;
        const bool rowvalid = (t0 + row) < nvalid;
        float wgt[QW];
#pragma unroll
        for (int j = 0; j < QW; ++j) {
          const float s = small[j * TT + t0 + row];
          const float a = __builtin_amdgcn_exp2f(fmaf(s, LOG2E, -mLq[j])) * il[j];
          wgt[j] = rowvalid ? a * (u[j] - dl[j]) : 0.f;
        }
        accumulate_rows<QW, KP>(wgt, xv, gacc);
      }
    }
    // per-IMAGE partial of sum_n dS x (reduced in image order + scaled by ep_reduce_partials)
#pragma unroll
    for (int j = 0; j < QW; ++j) {
      float* Gq = p.Gpart + ((int64_t)b_cur * Q + q0 + j) * D;
#pragma unroll
      for (int k = 0; k < KP; ++k) *reinterpret_cast<f4*>(Gq + 4 * (lane + 64 * k)) = gacc[j][k];
    }
    ++n_imgs;
    if (n_imgs == 1) stamp(wg, 2);
    if (b_next >= B) break;
    b_cur = b_next;
    b_next = B;
  }
  stamp(wg, 3);
  if (p.dbg && threadIdx.x == 0) { p.dbg[(int64_t)wg * 8 + 4] = (unsigned long long)n_imgs; p.dbg[(int64_t)wg * 8 + 5] = t_task; }
}

// ---------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------
template <int KP, int DFIX, bool BF16>
static int bwd2_launch_one(const PoolParams& p, int grid, int first, hipStream_t st, const SideTasks* side) {
  using Cfg = StreamCfgT<2, KP, 4>;
  const size_t slot = (size_t)Cfg::TT * p.D * 4;
  size_t lds = (size_t)Cfg::NSLOT_B * (slot + 4 * 256) + 64;           // ring + small pieces + the ticket word
  if (lds < ip_dp_lds_bytes(KP) + 64) lds = ip_dp_lds_bytes(KP) + 64;
  SideTasks sd{};
  if (side && side->total > 0) {
    sd = *side;
    if (lds < SIDE_LDS_BYTES) lds = SIDE_LDS_BYTES;
  }
  sd.first_block = first;
  auto k = ep_pool_bwd2_kernel<KP, DFIX, BF16>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e)); return (int)e; }
  static int stampmode = -1;            // diagnostic only (EP_IP_STAMP=1): per-workgroup timeline to stderr (synchronises!)
  if (stampmode < 0) { const char* ev = getenv("EP_IP_STAMP"); stampmode = ev ? atoi(ev) : 0; }
  if (stampmode && grid + sd.total <= 4096) {
    static unsigned long long* dbg = nullptr;
    const int nb = grid + sd.total;
    if (!dbg) (void)hipMalloc(&dbg, 4096 * 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg, 0, (size_t)nb * 8 * sizeof(unsigned long long), st);
    PoolParams q = p;
    q.dbg = dbg;
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), lds, st, q, sd);
    (void)hipStreamSynchronize(st);
    static unsigned long long host[4096 * 8];
    (void)hipMemcpy(host, dbg, (size_t)nb * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    static int printed = 0;
    if (printed++ % 20 == 10) {
      unsigned long long t0 = ~0ull;
      for (int g = 0; g < nb; ++g) if (host[g * 8] && host[g * 8] < t0) t0 = host[g * 8];
      auto us = [&](unsigned long long v) { return v ? (double)(v - t0) / 100.0 : -1.0; };
      fprintf(stderr, "[EP_IP_STAMP] bwd2: %d pooling workgroups (%d in front of %d side tasks); us since the first block started\n", grid, first, sd.total);
      for (int part = 0; part < 2; ++part) {
        const int g0 = part ? first : 0, g1 = part ? grid : first;
        if (g1 <= g0) continue;
        double m[4] = {0}, mx[4] = {0}, mn[4] = {1e9, 1e9, 1e9, 1e9}, imgs = 0, tt = 0; int cnt = 0;
        for (int g = g0; g < g1; ++g) {
          if (!host[g * 8 + 3]) continue;
          for (int k2 = 0; k2 < 4; ++k2) { const double v = us(host[g * 8 + k2]); m[k2] += v; if (v > mx[k2]) mx[k2] = v; if (v < mn[k2]) mn[k2] = v; }
          imgs += (double)host[g * 8 + 4]; tt += (double)host[g * 8 + 5] / 100.0; ++cnt;
        }
        if (!cnt) { fprintf(stderr, "   pooling wg %d..%d: none streamed an image\n", g0, g1 - 1); continue; }
        fprintf(stderr, "   pooling wg %4d..%4d (%d active): start %.1f [%.1f..%.1f]  stream-from %.1f [..%.1f]  first-image-done %.1f [%.1f..%.1f]  exit %.1f [%.1f..%.1f]  images/wg %.2f  task-point us/wg %.1f\n",
                g0, g1 - 1, cnt, m[0] / cnt, mn[0], mx[0], m[1] / cnt, mx[1], m[2] / cnt, mn[2], mx[2], m[3] / cnt, mn[3], mx[3], imgs / cnt, tt / cnt);
      }
      double s0 = 1e9, s1 = 0, e0 = 1e9, e1 = 0, dur = 0; int ns = 0;
      for (int g = grid; g < nb; ++g) {
        if (!host[g * 8 + 1]) continue;
        const double a = us(host[g * 8]), b = us(host[g * 8 + 1]);
        if (a < s0) s0 = a; if (a > s1) s1 = a; if (b < e0) e0 = b; if (b > e1) e1 = b; dur += b - a; ++ns;
      }
      if (ns) fprintf(stderr, "   side tasks (%d): starts %.1f..%.1f  ends %.1f..%.1f  mean duration %.1f\n", ns, s0, s1, e0, e1, dur / ns);
    }
    return 0;
  }
  hipLaunchKernelGGL(k, dim3(grid + sd.total), dim3(256), lds, st, p, sd);
  EP_LAUNCH_CHECK("ep_pool_bwd2_kernel");
  return 0;
}

// grid: pooling workgroups; first: how many of them are dispatched in front of the side tasks (they stream image `wg`
// first, everybody else draws a ticket)
int bwd2_launch(const PoolParams& p, int grid, int first, hipStream_t st, const SideTasks* side) {
  const int kp = p.D / 256;
  if (p.D == 768 && !p.x_bf16) return bwd2_launch_one<3, 768, false>(p, grid, first, st, side);
  if (p.D == 768) return bwd2_launch_one<3, 768, true>(p, grid, first, st, side);
  if (kp == 1) return p.x_bf16 ? bwd2_launch_one<1, 0, true>(p, grid, first, st, side) : bwd2_launch_one<1, 0, false>(p, grid, first, st, side);
  if (kp == 2) return p.x_bf16 ? bwd2_launch_one<2, 0, true>(p, grid, first, st, side) : bwd2_launch_one<2, 0, false>(p, grid, first, st, side);
  set_error("ticketed second pass: no kernel for D = %d", p.D);
  return EP_E_UNSUPPORTED;
}

}  // namespace ep
