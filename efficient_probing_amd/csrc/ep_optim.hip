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
};

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
  auto acc1 = [&](float pv, float graw) {
    const float gv = graw * o.inv_scale;
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

// update of one chunk: the body of ep_opt_update_kernel.  `partial`: the per-chunk sums (global memory, or the LDS copy of
// the small-segment kernel).
__device__ __forceinline__ void chunk_update(const OptParams& o, const OptSegs& segs, int chunk, int tid, float* sm,
                                             const float* partial, bool active) {
  const int k = seg_of_chunk(segs, chunk);
  const int64_t base = segs.off[k] + (int64_t)(chunk - segs.first_chunk[k]) * OPT_CHUNK;
  const int64_t end = segs.off[k] + segs.numel[k];
  // global non-finite flag and gradient norm (fixed order over all chunks)
  float bad = 0.f, gg = 0.f;
  for (int c = tid; c < o.nchunks; c += 256) { bad += partial[(int64_t)c * 4 + 3]; gg += partial[(int64_t)c * 4 + 2]; }
  // this thread's elements are fetched NOW (behind the first partial loads: vmcnt retires in order): the block sums and
  // the second chain of partial loads below run while they are in flight, instead of in front of them
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
  // the tensor's own sums (trust ratio) in the same round of loads and the same two barriers as the global ones
  const bool ratio = o.mode == 0 && segs.trust[k];
  float pp = 0.f, uu = 0.f;
  if (ratio)
    for (int c = segs.first_chunk[k] + tid; c < segs.first_chunk[k + 1]; c += 256) {
      pp += partial[(int64_t)c * 4 + 0]; uu += partial[(int64_t)c * 4 + 1];
    }
  const f4 sums = block_sum4(f4{bad, gg, pp, uu}, sm, tid);
  bad = sums.x; gg = sums.y; pp = sums.z; uu = sums.w;
  const bool aborted = o.abort_flag && *o.abort_flag != 0;       // uniform: every thread reads the same word
  if (aborted) bad += 1.f;
  if (active && chunk == 0 && tid == 0) {
    *o.found_inf = bad > 0.f ? 1 : 0;
    if (o.grad_norm) *o.grad_norm = sqrtf(gg);
    if (aborted && o.abort_stat) *o.abort_stat += 1.f;
  }
  const bool skip = bad > 0.f || !active;                // GradScaler.step: skip the update (block sums stay uniform)
  float q = 1.0f;
  if (ratio) {
    const float pn = sqrtf(pp), un = sqrtf(uu);
    q = (pn > 0.f && un > 0.f) ? o.tc * pn / un : 1.0f;  // util/lars.py:26-29
  }
  const bool decay = (o.mode == 0) ? (segs.trust[k] != 0) : (o.mode == 1);
  auto upd1 = [&](float pv, float graw, float& s0v, float& s1v) -> float {
    const float gv = graw * o.inv_scale;
    if (o.mode == 0) {                                   // LARS (util/lars.py:21-37)
      float dp = decay ? fmaf(o.wd, pv, gv) : gv;
      dp *= q;
      s0v = fmaf(s0v, o.momentum, dp);
      return pv - o.lr * s0v;
    } else if (o.mode == 1) {                            // SGD, no momentum (main_linprobe.py:407)
      const float d = (o.wd != 0.f) ? fmaf(o.wd, pv, gv) : gv;
      return pv - o.lr * d;
    } else {                                             // AdamW (torch defaults, decoupled decay)
      const float pw = pv * (1.0f - o.lr * o.wd);
      s0v = s0v * o.beta1 + (1.0f - o.beta1) * gv;
      s1v = s1v * o.beta2 + (1.0f - o.beta2) * gv * gv;
      const float denom = sqrtf(s1v) / o.bc2 + o.eps;    // bc2 = sqrt(1 - beta2^t)
      return pw - (o.lr / o.bc1) * (s0v / denom);        // bc1 = 1 - beta1^t
    }
  };
  if (skip) return;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int64_t i = base + tid * 4 + 1024 * v;
    if (i + 3 < end) {
      f4 pv = pvr[v];
      const f4 gv = gvr[v];
      float a[4] = {avr[v].x, avr[v].y, avr[v].z, avr[v].w}, b[4] = {bvr[v].x, bvr[v].y, bvr[v].z, bvr[v].w};
      pv.x = upd1(pv.x, gv.x, a[0], b[0]); pv.y = upd1(pv.y, gv.y, a[1], b[1]);
      pv.z = upd1(pv.z, gv.z, a[2], b[2]); pv.w = upd1(pv.w, gv.w, a[3], b[3]);
      *reinterpret_cast<f4*>(o.p + i) = pv;
      if (o.mode != 1) *reinterpret_cast<f4*>(o.s0 + i) = f4{a[0], a[1], a[2], a[3]};
      if (o.mode == 2) *reinterpret_cast<f4*>(o.s1 + i) = f4{b[0], b[1], b[2], b[3]};
    } else {
      for (int64_t t = i; t < end && t < i + 4; ++t) {
        float a = o.mode != 1 ? o.s0[t] : 0.f, b = o.mode == 2 ? o.s1[t] : 0.f;
        o.p[t] = upd1(o.p[t], o.g[t], a, b);
        if (o.mode != 1) o.s0[t] = a;
        if (o.mode == 2) o.s1[t] = b;
      }
    }
  }
}

__global__ __launch_bounds__(256) void ep_opt_update_kernel(OptParams o, OptSegs segs) {
  __shared__ __attribute__((aligned(16))) float sm[16];
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

size_t optim_workspace_bytes(int64_t total, int nseg) {
  return round_up(((size_t)(total / OPT_CHUNK) + (size_t)nseg + 1) * 4 * sizeof(float), 256);
}

int optim_step(int mode, float* p, const float* g, float* s0, float* s1, int64_t total, const ep_segment* segs,
               int nseg, float lr, float wd, float momentum, float tc, float inv_scale, float beta1, float beta2,
               float eps, int64_t step, int32_t* found_inf, float* grad_norm, void* ws, size_t ws_bytes,
               hipStream_t st, const DeferredReduce* red, const int* abort_flag, float* abort_stat) {
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
  hipLaunchKernelGGL(ep_opt_update_kernel, dim3(nchunks), dim3(256), 0, st, o, S);
  EP_LAUNCH_CHECK("ep_opt_update_kernel");
  return 0;
}

}  // namespace ep
