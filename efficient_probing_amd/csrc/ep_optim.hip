// Fused optimizer step over one flat fp32 parameter buffer (gfx950).
//
// Replaces reference util/lars.py:13-37 (LARS), torch.optim.SGD / AdamW as selected in
// main_linprobe.py:403-408, the GradScaler unscale + inf/nan skip of util/misc.py:267-277 and
// get_grad_norm_ (util/misc.py:289-301).  Two launches:
//   1. ep_opt_norms_kernel : per-chunk partial sums  ||p||^2, ||g*inv + wd*p||^2, ||g*inv||^2 and
//      a non-finite flag (one pass over p and g);
//   2. ep_opt_update_kernel: every chunk re-reduces the partials of its tensor in a fixed order
//      (deterministic), forms the trust ratio, checks the global non-finite flag and updates
//      p / state in place (one pass over p, g, state).
#include "ep_common.h"
#include "ep_internal.h"
#include "ep_planes_dev.h"

namespace ep {

constexpr int OPT_CHUNK = 4096;      // elements per workgroup
constexpr int OPT_MAX_SEG = 32;

struct OptSegs {
  int n;
  int64_t off[OPT_MAX_SEG];
  int64_t numel[OPT_MAX_SEG];
  int first_chunk[OPT_MAX_SEG + 1];   // chunk index range of each segment
  int trust[OPT_MAX_SEG];
};

struct OptParams {
  float* p; const float* g; float* s0; float* s1;
  float lr, wd, momentum, tc, inv_scale, beta1, beta2, eps, bc1, bc2;
  int mode;                           // 0 LARS, 1 SGD, 2 AdamW
  float* partial;                     // [nchunks][4]
  int nchunks;
  int32_t* found_inf; float* grad_norm;
  // deferred last stage of a partial reduction (DeferredReduce): g[red_off .. red_off + red_n) is produced HERE, by the
  // norms kernel, from 16 stage rows -- and written back to g for the update kernel and for readers of the gradients
  const float* red_stage; float* gw; int64_t red_off; int red_n; float red_scale; int red_accumulate;
  // a nonzero *abort_flag (the give-up count of the in-pass hand-off waits, ep_inpass.h) makes this step count as
  // non-finite: the update is skipped, found_inf is set and *abort_stat (the step statistics' non-finite row count, which
  // stops the training loop) is bumped -- a pass that read unfinished rows must never reach the parameters silently
  const int* abort_flag; float* abort_stat;
  // many chunks (> OPT_FINAL_MIN): the per-chunk partials are reduced ONCE by ep_opt_finalize_kernel into
  // final_[0..1] = {non-finite count, sum g^2}, final_[2 + 2k .. 3 + 2k] = {sum p^2, sum u^2} of segment k -- in the order
  // opt_scalars sums them -- instead of by every workgroup of the update (5094 chunks at 196 x 4096 tokens: 80 KB of
  // partials re-read from L2 per 64 KB of payload, and a serial latency chain in front of every workgroup's stores)
  const float* final_;
  // device-resident loss scale (ScalerDev): gradients are unscaled by inv_scale / *sc_in; the thread that publishes found_inf
  // also writes the NEXT state {scale, tracker} to sc_out (torch.cuda.amp.GradScaler.update)
  const float* sc_in; float* sc_out; float sc_growth, sc_backoff; int sc_interval;
};
__device__ __forceinline__ float opt_inv_scale(const OptParams& o) { return o.sc_in ? o.inv_scale / o.sc_in[0] : o.inv_scale; }
constexpr int OPT_FINAL_MIN = 1024;

__device__ __forceinline__ int seg_of_chunk(const OptSegs& s, int chunk) {
  int k = 0;
  while (k + 1 < s.n && chunk >= s.first_chunk[k + 1]) ++k;
  return k;
}

// sum over the 256 threads of a GROUP (a whole 256-thread workgroup, or one quarter of the 1024-thread small-segment
// kernel: `tid` is the index inside the group, `sm` the group's four floats); every thread of the workgroup must call it
__device__ __forceinline__ float block_sum(float v, float* sm, int tid) {
  v = wave_sum(v);
  const int w = tid >> 6;
  if ((tid & 63) == 0) sm[w] = v;
  __syncthreads();
  float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
}

// four such sums at once (two barriers instead of eight): `sm` holds 16 floats here.  Component by component the same
// arithmetic as block_sum -- wave sums, then (w0 + w1) + (w2 + w3).
__device__ __forceinline__ f4 block_sum4(f4 v, float* sm, int tid) {
  v.x = wave_sum(v.x); v.y = wave_sum(v.y); v.z = wave_sum(v.z); v.w = wave_sum(v.w);
  const int w = tid >> 6;
  if ((tid & 63) == 0) *reinterpret_cast<f4*>(sm + 4 * w) = v;
  __syncthreads();
  const f4 a = *reinterpret_cast<const f4*>(sm), b = *reinterpret_cast<const f4*>(sm + 4), c = *reinterpret_cast<const f4*>(sm + 8),
           d = *reinterpret_cast<const f4*>(sm + 12);
  const f4 r = (a + b) + (c + d);
  __syncthreads();
  return r;
}

// norms of one chunk: the body of ep_opt_norms_kernel.  Returns the four partial sums in every thread of the group.
__device__ __forceinline__ f4 chunk_norms(const OptParams& o, const OptSegs& segs, int chunk, int tid, float* sm) {
  const int k = seg_of_chunk(segs, chunk);
  const int64_t base = segs.off[k] + (int64_t)(chunk - segs.first_chunk[k]) * OPT_CHUNK;
  const int64_t end = segs.off[k] + segs.numel[k];
  const bool decay = (o.mode == 0) ? (segs.trust[k] != 0) : (o.mode == 1);
  float pp = 0.f, uu = 0.f, gg = 0.f, bad = 0.f;
  const float inv_scale = opt_inv_scale(o);
  auto acc1 = [&](float pv, float graw) {
    const float gv = graw * inv_scale;
    const float u = decay ? fmaf(o.wd, pv, gv) : gv;
    pp = fmaf(pv, pv, pp); uu = fmaf(u, u, uu); gg = fmaf(gv, gv, gg);
    bad += (fabsf(gv) <= 3.4028234664e38f) ? 0.f : 1.f;
  };
  // a chunk of the deferred reduction's range (optim_step: the range is a float4-aligned part of ONE segment, so a chunk
  // is inside it or outside): its gradients are summed here from the 16 stage rows, in ep_reduce_partials_kernel's order
  const bool redchunk = o.red_stage && base >= o.red_off && base < o.red_off + o.red_n;
  if (redchunk) {
    const int64_t rend = (o.red_off + o.red_n) < end ? (o.red_off + o.red_n) : end;
#pragma unroll
    for (int v = 0; v < OPT_CHUNK / 1024; ++v) {                  // straight-line loads (clamped), guarded uses
      const int64_t i = base + tid * 4 + 1024 * v;
      const bool in = i + 3 < rend;
      const int64_t ic = in ? i : base;
      const float* sp = o.red_stage + (ic - o.red_off);
      f4 r[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) r[u] = *reinterpret_cast<const f4*>(sp + (int64_t)u * o.red_n);
      const f4 pv = *reinterpret_cast<const f4*>(o.p + ic);
      const f4 old = *reinterpret_cast<const f4*>(o.g + ic);
      f4 t[4];
#pragma unroll
      for (int py = 0; py < 4; ++py) t[py] = (r[py] + r[py + 8]) + (r[py + 4] + r[py + 12]);
      const f4 ts = (t[0] + t[1]) + (t[2] + t[3]);
      f4 gv = ts * o.red_scale;
      if (o.red_accumulate)                 // one rounding, as in ep_reduce_partials_kernel
        gv = f4{fmaf(ts.x, o.red_scale, old.x), fmaf(ts.y, o.red_scale, old.y), fmaf(ts.z, o.red_scale, old.z), fmaf(ts.w, o.red_scale, old.w)};
      if (in) {
        if (o.gw) *reinterpret_cast<f4*>(o.gw + i) = gv;        // (null: a shadow group of the small-segment kernel)
        acc1(pv.x, gv.x); acc1(pv.y, gv.y); acc1(pv.z, gv.z); acc1(pv.w, gv.w);
      } else if (i < end) {                                        // elements of the segment behind the range
        for (int64_t t = i; t < end && t < i + 4; ++t) acc1(o.p[t], o.g[t]);
      }
    }
  } else
  for (int e = tid * 4; e < OPT_CHUNK; e += 1024) {     // segment offsets are multiples of 4
    const int64_t i = base + e;
    if (i + 3 < end) {
      const f4 pv = *reinterpret_cast<const f4*>(o.p + i);
      const f4 gv = *reinterpret_cast<const f4*>(o.g + i);
      acc1(pv.x, gv.x); acc1(pv.y, gv.y); acc1(pv.z, gv.z); acc1(pv.w, gv.w);
    } else {
      for (int64_t t = i; t < end && t < i + 4; ++t) acc1(o.p[t], o.g[t]);
    }
  }
  return block_sum4(f4{pp, uu, gg, bad}, sm, tid);
}

__global__ __launch_bounds__(256) void ep_opt_norms_kernel(OptParams o, OptSegs segs) {
  __shared__ __attribute__((aligned(16))) float sm[16];
  const f4 r = chunk_norms(o, segs, blockIdx.x, threadIdx.x, sm);
  if (threadIdx.x == 0) *reinterpret_cast<f4*>(o.partial + (int64_t)blockIdx.x * 4) = r;
}

// one element's update: LARS (util/lars.py:21-37), SGD without momentum (main_linprobe.py:407), AdamW (torch defaults)
__device__ __forceinline__ float opt_update_elem(const OptParams& o, bool decay, float q, float pv, float graw, float& s0v, float& s1v) {
  const float gv = graw * o.inv_scale;          // (callers with a device-resident loss scale pass a copy of `o` whose inv_scale is resolved)
  if (o.mode == 0) {
    float dp = decay ? fmaf(o.wd, pv, gv) : gv;
    dp *= q;
    s0v = fmaf(s0v, o.momentum, dp);
    return pv - o.lr * s0v;
  } else if (o.mode == 1) {
    const float d = (o.wd != 0.f) ? fmaf(o.wd, pv, gv) : gv;
    return pv - o.lr * d;
  } else {
    const float pw = pv * (1.0f - o.lr * o.wd);
    s0v = s0v * o.beta1 + (1.0f - o.beta1) * gv;
    s1v = s1v * o.beta2 + (1.0f - o.beta2) * gv * gv;
    const float denom = sqrtf(s1v) / o.bc2 + o.eps;    // bc2 = sqrt(1 - beta2^t)
    return pw - (o.lr / o.bc1) * (s0v / denom);        // bc1 = 1 - beta1^t
  }
}

// The scalars every workgroup of the update needs, from the per-chunk partial sums in a fixed order: the global non-finite
// flag and gradient norm, and tensor k's trust ratio.  All 256 threads of the group call it; `first` (one thread of the
// launch) also publishes found_inf / grad_norm.  Returns {skip, q}.
struct OptScalars { bool skip; float q; };
__device__ __forceinline__ OptScalars opt_scalars(const OptParams& o, const OptSegs& segs, int k, int tid, float* sm,
                                                  const float* partial, bool active, bool first) {
  float bad = 0.f, gg = 0.f;
  const bool ratio = o.mode == 0 && segs.trust[k];
  float pp = 0.f, uu = 0.f;
  if (o.final_ && partial == o.partial) {            // already reduced (same order, same bits): four loads
    bad = o.final_[0]; gg = o.final_[1];
    if (ratio) { pp = o.final_[2 + 2 * k]; uu = o.final_[3 + 2 * k]; }
  } else {
    for (int c = tid; c < o.nchunks; c += 256) { bad += partial[(int64_t)c * 4 + 3]; gg += partial[(int64_t)c * 4 + 2]; }
    if (ratio)
      for (int c = segs.first_chunk[k] + tid; c < segs.first_chunk[k + 1]; c += 256) {
        pp += partial[(int64_t)c * 4 + 0]; uu += partial[(int64_t)c * 4 + 1];
      }
    const f4 sums = block_sum4(f4{bad, gg, pp, uu}, sm, tid);
    bad = sums.x; gg = sums.y; pp = sums.z; uu = sums.w;
  }
  const bool aborted = o.abort_flag && *o.abort_flag != 0;       // uniform: every thread reads the same word
  if (aborted) bad += 1.f;
  if (active && first) {
    *o.found_inf = bad > 0.f ? 1 : 0;
    if (o.grad_norm) *o.grad_norm = sqrtf(gg);
    if (aborted && o.abort_stat) *o.abort_stat += 1.f;
    if (o.sc_out) {                                   // GradScaler.update(): the next step reads this slot
      float sc = o.sc_in[0], tr = o.sc_in[1];
      if (bad > 0.f) { sc *= o.sc_backoff; tr = 0.f; }
      else { tr += 1.f; if (tr >= (float)o.sc_interval) { sc *= o.sc_growth; tr = 0.f; } }
      o.sc_out[0] = sc; o.sc_out[1] = tr;
    }
  }
  float q = 1.0f;
  if (ratio) {
    const float pn = sqrtf(pp), un = sqrtf(uu);
    q = (pn > 0.f && un > 0.f) ? o.tc * pn / un : 1.0f;  // util/lars.py:26-29
  }
  return OptScalars{bad > 0.f || !active, q};            // GradScaler.step: skip the update (block sums stay uniform)
}

// update of one chunk: the body of ep_opt_update_kernel.  `partial`: the per-chunk sums (global memory, or the LDS copy of
// the small-segment kernel).
__device__ __forceinline__ void chunk_update(const OptParams& o_, const OptSegs& segs, int chunk, int tid, float* sm,
                                             const float* partial, bool active) {
  OptParams o = o_;
  o.inv_scale = opt_inv_scale(o_);
  const int k = seg_of_chunk(segs, chunk);
  const int64_t base = segs.off[k] + (int64_t)(chunk - segs.first_chunk[k]) * OPT_CHUNK;
  const int64_t end = segs.off[k] + segs.numel[k];
  // this thread's elements are fetched NOW, in front of the partial loads' reductions: the block sums run while they are
  // in flight, instead of in front of them
  constexpr int NV = OPT_CHUNK / 1024;
  f4 pvr[NV], gvr[NV], avr[NV], bvr[NV];
  // branch-free (a conditional load makes the compiler wait for everything in flight at the join): unused state buffers
  // alias the parameters, float4 groups past the segment end re-read its first one
  const float* s0p = o.mode != 1 ? o.s0 : o.p;
  const float* s1p = o.mode == 2 ? o.s1 : o.p;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int64_t i = base + tid * 4 + 1024 * v;
    const int64_t ic = (i + 3 < end) ? i : base;
    pvr[v] = *reinterpret_cast<const f4*>(o.p + ic);
    gvr[v] = *reinterpret_cast<const f4*>(o.g + ic);
    avr[v] = *reinterpret_cast<const f4*>(s0p + ic);
    bvr[v] = *reinterpret_cast<const f4*>(s1p + ic);
  }
  const OptScalars sc = opt_scalars(o, segs, k, tid, sm, partial, active, chunk == 0 && tid == 0);
  const float q = sc.q;
  const bool decay = (o.mode == 0) ? (segs.trust[k] != 0) : (o.mode == 1);
  if (sc.skip) return;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int64_t i = base + tid * 4 + 1024 * v;
    if (i + 3 < end) {
      f4 pv = pvr[v];
      const f4 gv = gvr[v];
      float a[4] = {avr[v].x, avr[v].y, avr[v].z, avr[v].w}, b[4] = {bvr[v].x, bvr[v].y, bvr[v].z, bvr[v].w};
      pv.x = opt_update_elem(o, decay, q, pv.x, gv.x, a[0], b[0]); pv.y = opt_update_elem(o, decay, q, pv.y, gv.y, a[1], b[1]);
      pv.z = opt_update_elem(o, decay, q, pv.z, gv.z, a[2], b[2]); pv.w = opt_update_elem(o, decay, q, pv.w, gv.w, a[3], b[3]);
      *reinterpret_cast<f4*>(o.p + i) = pv;
      if (o.mode != 1) *reinterpret_cast<f4*>(o.s0 + i) = f4{a[0], a[1], a[2], a[3]};
      if (o.mode == 2) *reinterpret_cast<f4*>(o.s1 + i) = f4{b[0], b[1], b[2], b[3]};
    } else {
      for (int64_t t = i; t < end && t < i + 4; ++t) {
        float a = o.mode != 1 ? o.s0[t] : 0.f, b = o.mode == 2 ? o.s1[t] : 0.f;
        o.p[t] = opt_update_elem(o, decay, q, o.p[t], o.g[t], a, b);
        if (o.mode != 1) o.s0[t] = a;
        if (o.mode == 2) o.s1[t] = b;
      }
    }
  }
}

// block 0: the global sums; block 1 + k: segment k's.  Thread tid takes chunks tid, tid + 256, ... and the block sums
// follow -- component by component the arithmetic of opt_scalars, so an update that reads these gets the bits it would
// have computed itself.
__global__ __launch_bounds__(256) void ep_opt_finalize_kernel(OptParams o, OptSegs segs, float* fin) {
  __shared__ __attribute__((aligned(16))) float sm[16];
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    float bad = 0.f, gg = 0.f;
    for (int c = tid; c < o.nchunks; c += 256) { bad += o.partial[(int64_t)c * 4 + 3]; gg += o.partial[(int64_t)c * 4 + 2]; }
    const f4 r = block_sum4(f4{bad, gg, 0.f, 0.f}, sm, tid);
    if (tid == 0) { fin[0] = r.x; fin[1] = r.y; }
  } else {
    const int k = blockIdx.x - 1;
    float pp = 0.f, uu = 0.f;
    for (int c = segs.first_chunk[k] + tid; c < segs.first_chunk[k + 1]; c += 256) {
      pp += o.partial[(int64_t)c * 4 + 0]; uu += o.partial[(int64_t)c * 4 + 1];
    }
    const f4 r = block_sum4(f4{0.f, 0.f, pp, uu}, sm, tid);
    if (tid == 0) { fin[2 + 2 * k] = r.z; fin[3 + 2 * k] = r.w; }
  }
}

// Weight matrices whose bf16 planes (ep_planes.hip: the operands of the bf16 x3 contractions of the NEXT step) the update
// writes as it updates them: such a segment is walked in 64 x 64 TILES by extra workgroups of the update launch -- parameters,
// gradient and state read once, the update applied element by element exactly as chunk_update does, the new parameters
// staged in LDS and written out as planes of both orientations -- instead of a split launch of 268 MB of traffic per step
// beside the first token pass (196 x 4096 tokens: 3 x 236 us of ep_planes_split_kernel, the pass slowed from 582 to 672 us).
struct PlaneEmit {
  PlaneJob j[2]; int seg[2]; int tiles_x[2]; int first_block[3];     // blocks [first_block[i], first_block[i+1]) walk job i
  int n;
};

__device__ __forceinline__ void tile_update_emit(const OptParams& o_, const OptSegs& segs, const PlaneEmit& pe, int job, int t,
                                                 int tid, float* sm, float (*tile)[65]) {
  OptParams o = o_;
  o.inv_scale = opt_inv_scale(o_);
  const PlaneJob& jb = pe.j[job];
  const int k = pe.seg[job];
  const int r0 = (t / pe.tiles_x[job]) * 64, c0 = (t % pe.tiles_x[job]) * 64;
  float* P = o.p + segs.off[k];
  const float* G = o.g + segs.off[k];
  float* S0 = (o.mode != 1 ? o.s0 : o.p) + segs.off[k];
  float* S1 = (o.mode == 2 ? o.s1 : o.p) + segs.off[k];
  // this thread's 4 x float4 (row r0 + (tid >> 4) + 16 i, columns c0 + 4 (tid & 15) ...): fetched in front of the block
  // sums, branch-free (clamped address, masked use).  K is a multiple of 4 (host-checked), so a float4 is inside the matrix
  // or outside it as a whole.
  f4 pv[4], gv[4], av[4], bv[4];
  int64_t at[4]; bool in[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + (tid >> 4) + 16 * i, c = c0 + 4 * (tid & 15);
    in[i] = r < jb.R && c < jb.K;
    at[i] = in[i] ? (int64_t)r * jb.ldw + c : 0;
    pv[i] = *reinterpret_cast<const f4*>(P + at[i]); gv[i] = *reinterpret_cast<const f4*>(G + at[i]);
    av[i] = *reinterpret_cast<const f4*>(S0 + at[i]); bv[i] = *reinterpret_cast<const f4*>(S1 + at[i]);
  }
  const OptScalars sc = opt_scalars(o, segs, k, tid, sm, o.partial, true, false);
  if (sc.skip) return;                               // parameters unchanged: the planes of the last update still hold
  const bool decay = (o.mode == 0) ? (segs.trust[k] != 0) : (o.mode == 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a[4] = {av[i].x, av[i].y, av[i].z, av[i].w}, b[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
    f4 nv;
    nv.x = opt_update_elem(o, decay, sc.q, pv[i].x, gv[i].x, a[0], b[0]); nv.y = opt_update_elem(o, decay, sc.q, pv[i].y, gv[i].y, a[1], b[1]);
    nv.z = opt_update_elem(o, decay, sc.q, pv[i].z, gv[i].z, a[2], b[2]); nv.w = opt_update_elem(o, decay, sc.q, pv[i].w, gv[i].w, a[3], b[3]);
    float* trow = &tile[(tid >> 4) + 16 * i][4 * (tid & 15)];
    trow[0] = in[i] ? nv.x : 0.f; trow[1] = in[i] ? nv.y : 0.f; trow[2] = in[i] ? nv.z : 0.f; trow[3] = in[i] ? nv.w : 0.f;
    if (in[i]) {
      *reinterpret_cast<f4*>(P + at[i]) = nv;
      if (o.mode != 1) *reinterpret_cast<f4*>(S0 + at[i]) = f4{a[0], a[1], a[2], a[3]};
      if (o.mode == 2) *reinterpret_cast<f4*>(S1 + at[i]) = f4{b[0], b[1], b[2], b[3]};
    }
  }
  __syncthreads();
  pl_emit_tile(jb, tile, r0, c0, tid);
}

__global__ __launch_bounds__(256) void ep_opt_update_kernel(OptParams o, OptSegs segs, PlaneEmit pe) {
  __shared__ __attribute__((aligned(16))) float sm[16];
  if (pe.n > 0) {
    __shared__ float tile[64][65];
    const int blk = blockIdx.x;
    if (blk >= pe.first_block[0]) {
      const int job = (pe.n > 1 && blk >= pe.first_block[1]) ? 1 : 0;
      tile_update_emit(o, segs, pe, job, blk - pe.first_block[job], threadIdx.x, sm, tile);
      return;
    }
    const int k = seg_of_chunk(segs, blk);             // chunks of a tile-walked segment: nothing to do here
    if (k == pe.seg[0] || (pe.n > 1 && k == pe.seg[1])) {
      // ... except that chunk 0 still publishes found_inf / grad_norm (the whole workgroup runs the block sums)
      if (blk == 0) (void)opt_scalars(o, segs, k, threadIdx.x, sm, o.partial, true, threadIdx.x == 0);
      return;
    }
  }
  chunk_update(o, segs, blockIdx.x, threadIdx.x, sm, o.partial, true);
}

// One SMALL range of segments (at most OPT_SMALL_CHUNKS chunks: the cls_token of the EP head is 6144 elements = 2 chunks)
// in ONE launch of one workgroup: group g of 256 threads runs chunk g's norms body, the partials go through LDS, the same
// group then runs chunk g's update body.  Every sum is taken exactly as the two-launch path takes it (bit-equal results);
// what goes away is one launch boundary and the round trip of the partials through memory -- the deferred-update step
// (ep_head_train_step, phases bit 4) has only this in front of the next step's first token pass.
constexpr int OPT_SMALL_CHUNKS = 4;
__global__ __launch_bounds__(256 * OPT_SMALL_CHUNKS) void ep_opt_small_kernel(OptParams o, OptSegs segs) {
  __shared__ __attribute__((aligned(16))) float sm[OPT_SMALL_CHUNKS][16];
  __shared__ __attribute__((aligned(16))) float part[OPT_SMALL_CHUNKS * 4];
  const int g = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const bool active = g < o.nchunks;
  const int chunk = active ? g : o.nchunks - 1;          // idle groups shadow the last chunk (barriers stay uniform), write nothing
  OptParams on = o;
  if (!active) on.gw = nullptr;
  const f4 r = chunk_norms(on, segs, chunk, tid, sm[g]);
  if (active && tid == 0) *reinterpret_cast<f4*>(part + 4 * g) = r;
  __syncthreads();
  chunk_update(o, segs, chunk, tid, sm[g], part, active);
}


static int build_segs(const ep_segment* segs, int nseg, int64_t total, OptSegs& out, int& nchunks) {
  EP_REQUIRE(nseg >= 1 && nseg <= OPT_MAX_SEG, EP_E_ARG, "optimizer: 1..%d segments supported, got %d", OPT_MAX_SEG, nseg);
  out.n = nseg;
  int c = 0;
  for (int k = 0; k < nseg; ++k) {
    EP_REQUIRE(segs[k].offset >= 0 && segs[k].numel > 0 && segs[k].offset + segs[k].numel <= total, EP_E_ARG,
               "optimizer: segment %d out of range", k);
    out.off[k] = segs[k].offset; out.numel[k] = segs[k].numel; out.trust[k] = segs[k].apply_trust;
    out.first_chunk[k] = c;
    c += (int)((segs[k].numel + OPT_CHUNK - 1) / OPT_CHUNK);
  }
  out.first_chunk[nseg] = c;
  nchunks = c;
  return 0;
}

int optim_emits(const float* p, const ep_segment* segs, int nseg, const PlaneSpec* emit, int n_emit) {
  int n = 0;
  for (int i = 0; i < n_emit && emit && n < 2; ++i)
    for (int t = 0; t < nseg; ++t)
      if (p + segs[t].offset == emit[i].W && segs[t].numel == (int64_t)emit[i].R * emit[i].K && emit[i].ldw == emit[i].K &&
          (emit[i].pn || emit[i].pt)) { ++n; break; }
  return n;
}

size_t optim_workspace_bytes(int64_t total, int nseg) {
  // per-chunk partials (every segment rounds its chunk count up) + the reduced sums of ep_opt_finalize_kernel
  return round_up(((size_t)(total / OPT_CHUNK) + (size_t)nseg + 1) * 4 * sizeof(float) + (2 + 2 * (size_t)nseg) * sizeof(float), 256);
}

int optim_step(int mode, float* p, const float* g, float* s0, float* s1, int64_t total, const ep_segment* segs,
               int nseg, float lr, float wd, float momentum, float tc, float inv_scale, float beta1, float beta2,
               float eps, int64_t step, int32_t* found_inf, float* grad_norm, void* ws, size_t ws_bytes,
               hipStream_t st, const DeferredReduce* red, const int* abort_flag, float* abort_stat,
               const PlaneSpec* emit, int n_emit, const ScalerDev* scaler) {
  EP_REQUIRE(p && g && found_inf && ws, EP_E_ARG, "optimizer: null pointer");
  EP_REQUIRE(mode != 0 || s0, EP_E_ARG, "LARS needs the momentum buffer");
  EP_REQUIRE(mode != 2 || (s0 && s1), EP_E_ARG, "AdamW needs exp_avg and exp_avg_sq");
  OptSegs S{};
  int nchunks = 0;
  ep_segment whole{0, total, 0, 0};
  if (!segs) { segs = &whole; nseg = 1; }
  EP_TRY(build_segs(segs, nseg, total, S, nchunks));
  EP_REQUIRE(ws_bytes >= (size_t)nchunks * 4 * sizeof(float), EP_E_WORKSPACE, "optimizer workspace too small");
  OptParams o{};
  o.p = p; o.g = g; o.s0 = s0; o.s1 = s1; o.lr = lr; o.wd = wd; o.momentum = momentum; o.tc = tc;
  o.inv_scale = inv_scale; o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.mode = mode;
  if (mode == 2) {
    o.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    o.bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  }
  o.partial = (float*)ws; o.nchunks = nchunks; o.found_inf = found_inf; o.grad_norm = grad_norm;
  o.abort_flag = abort_flag; o.abort_stat = abort_stat;
  if (scaler && scaler->state) {
    EP_REQUIRE(scaler->slot == 0 || scaler->slot == 1, EP_E_ARG, "optimizer: scaler slot %d", scaler->slot);
    o.sc_in = scaler->state + 2 * scaler->slot; o.sc_out = scaler->state + 2 * (1 - scaler->slot);
    o.sc_growth = scaler->growth; o.sc_backoff = scaler->backoff; o.sc_interval = scaler->interval;
  }
  if (red && red->stage) {
    // the range must be one whole segment of whole float4s (the norms kernel decides per chunk)
    const int64_t off = red->out - g;
    bool inside = false;
    for (int k = 0; k < nseg; ++k)
      inside |= off == segs[k].offset && red->n == segs[k].numel;     // the range IS a segment: chunks are inside it or outside
    if (inside && off >= 0 && off % 4 == 0 && red->n % 4 == 0 && aligned16(red->stage) && aligned16(g)) {
      o.red_stage = red->stage; o.gw = const_cast<float*>(g); o.red_off = off; o.red_n = red->n;
      o.red_scale = red->scale; o.red_accumulate = red->accumulate;
    } else {
      EP_TRY(reduce_partials(red->stage, 16, red->n, red->scale, red->accumulate, red->out, nullptr, st));
    }
  }
  static int small_on = -1;
  if (small_on < 0) { const char* e = getenv("EP_OPT_SMALL"); small_on = e ? atoi(e) : 0; }   // off: 17 us in one workgroup against 7 + 8 us for the two launches (same latency chain)
  if (small_on && nseg == 1 && nchunks <= OPT_SMALL_CHUNKS) {
    // one small tensor (the cls_token of a split / deferred update): norms and update in ONE launch, same sums
    hipLaunchKernelGGL(ep_opt_small_kernel, dim3(1), dim3(256 * OPT_SMALL_CHUNKS), 0, st, o, S);
    EP_LAUNCH_CHECK("ep_opt_small_kernel");
    return 0;
  }
  hipLaunchKernelGGL(ep_opt_norms_kernel, dim3(nchunks), dim3(256), 0, st, o, S);
  EP_LAUNCH_CHECK("ep_opt_norms_kernel");
  static int fin_min = -1;
  if (fin_min < 0) { const char* e = getenv("EP_OPT_FINAL_MIN"); fin_min = e ? atoi(e) : OPT_FINAL_MIN; }
  if (nchunks > fin_min && ws_bytes >= ((size_t)nchunks * 4 + 2 + 2 * (size_t)nseg) * sizeof(float)) {
    float* fin = o.partial + (size_t)nchunks * 4;      // the slack optim_workspace_bytes leaves behind the partials
    hipLaunchKernelGGL(ep_opt_finalize_kernel, dim3(1 + nseg), dim3(256), 0, st, o, S, fin);
    EP_LAUNCH_CHECK("ep_opt_finalize_kernel");
    o.final_ = fin;
  }
  // weight matrices whose planes this update writes (ep_planes.hip): each must be exactly one of the segments
  PlaneEmit pe{};
  int grid = nchunks;
  for (int i = 0; i < n_emit && emit && pe.n < 2; ++i) {
    const PlaneSpec& sp = emit[i];
    int k = -1;
    for (int t = 0; t < nseg; ++t)
      if (p + segs[t].offset == sp.W && segs[t].numel == (int64_t)sp.R * sp.K && sp.ldw == sp.K) k = t;
    if (k < 0 || !(sp.pn || sp.pt)) continue;          // not among the tensors of this call: the caller splits it itself
    EP_REQUIRE(sp.K % 4 == 0 && aligned16(sp.W) && aligned16(g + segs[k].offset) && (mode == 1 || aligned16(s0 + segs[k].offset)),
               EP_E_ALIGN, "optimizer: a matrix whose planes the update writes needs K %% 4 == 0 and 16-byte aligned tensors");
    PlaneJob& j = pe.j[pe.n];
    j.W = sp.W; j.R = sp.R; j.K = sp.K; j.ldw = sp.ldw;
    j.pn = sp.pn; j.pn_ld = (int64_t)round_up((size_t)sp.K, 32); j.pn_term = (int64_t)sp.R * j.pn_ld;
    j.pt = sp.pt; j.pt_ld = (int64_t)round_up((size_t)sp.R, 32); j.pt_term = (int64_t)sp.K * j.pt_ld;
    pe.seg[pe.n] = k; pe.tiles_x[pe.n] = (sp.K + 63) / 64;
    pe.first_block[pe.n] = grid;
    grid += pe.tiles_x[pe.n] * ((sp.R + 63) / 64);
    pe.first_block[pe.n + 1] = grid;
    ++pe.n;
  }
  hipLaunchKernelGGL(ep_opt_update_kernel, dim3(grid), dim3(256), 0, st, o, S, pe);
  EP_LAUNCH_CHECK("ep_opt_update_kernel");
  return 0;
}

}  // namespace ep
