/* ep_hip.h -- C ABI of the MI355X-native efficient-probing (EP) head engine.
 *
 * Drop-in boundary for the hot path of billpsomas/efficient-probing.  The reference has no
 * FFI of its own: its boundary is the Python module protocol
 *     model.head = Sequential(pooling, BatchNorm1d(affine=False, eps=1e-6), Linear)
 * (reference probe_heads.py:87-110) driven by engine_finetune.py:22-103 and util/lars.py.
 * Each entry point below replaces the torch ops behind one piece of that protocol; the
 * reference file:line it replaces is cited on every declaration.  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions (all entry points):
 *   - every pointer is a CALLER-OWNED DEVICE buffer (hipMalloc / torch tensor storage); the
 *     library never allocates persistent memory and never frees caller memory.  Scratch is
 *     passed in explicitly; its size comes from the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call returns
 *     without synchronising the device.  A call can be captured into a hipGraph (round 6:
 *     tests/test_gpu_graph_capture.py captures whole ep_head_train_step calls and replays them bit-identically), with
 *     two things to know: the FIRST call of a kind must run eagerly (it sets kernel attributes and creates the library's
 *     events), and scalars are passed by VALUE -- a captured step replays with ITS lr / opt_step / planes_valid, so a
 *     learning-rate schedule needs a re-capture per value (the loss scale is the exception: ep_head_step.scaler_state
 *     lives on the device).  Replays measure the same device time as eager steps (the step is GPU-bound).
 *   - return value: 0 = ok; negative = invalid argument / unsupported shape (EP_E_*);
 *     positive = hipError_t of a failed launch.  Nothing throws across the boundary.
 *     ep_last_error_string() returns a thread-local description of the last failure.
 *   - stateless and re-entrant: distinct host threads may call concurrently on distinct streams.
 *   - all tensors are row-major float32 unless a dtype argument says otherwise; token tensors
 *     `x` are (B, N, D) with the last two dimensions contiguous and an explicit batch stride
 *     (in elements), so the `feat[:, 1:]` patch-token view of reference models_more.py:24 is
 *     accepted without a copy.
 */
#ifndef EP_HIP_H
#define EP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EP_ABI_VERSION 26

#define EP_DTYPE_F32 0
#define EP_DTYPE_BF16 1
#define EP_DTYPE_F16 2        /* ABI v24: fp16-STORED tokens, FORWARD entry points only (ep_pool_forward, ep_head_eval_forward): what the
                              reference's evaluate() hands the head under its fp16 autocast (engine_finetune.py:131); widened
                              to fp32 in the token ring like bf16 -- exact -- and all arithmetic stays fp32.  Every other
                              entry point rejects it (EP_E_UNSUPPORTED). */
#define EP_ARITH_F32 0            /* ep_head_step.arith (ABI v25) */
#define EP_ARITH_BF16_AUTOCAST 1

#define EP_OK 0
#define EP_E_ARG (-1)         /* null pointer / non-positive size / bad enum            */
#define EP_E_SHAPE (-2)       /* shape constraint violated (e.g. D % (d_out*Q) != 0)     */
#define EP_E_ALIGN (-3)       /* pointer or stride not 16-byte aligned where required    */
#define EP_E_WORKSPACE (-4)   /* workspace too small                                     */
#define EP_E_UNSUPPORTED (-5) /* dtype / mode not implemented                            */

typedef void* ep_stream_t;

int ep_version(void);
const char* ep_last_error_string(void);
/* number of compute units of the current device (used by callers to size persistent grids) */
int ep_device_cu_count(void);
/* test hook: select the pooling kernel family so the independent implementations can be compared
 * on identical inputs: 0 = automatic (matrix-core streaming kernel where supported, else the
 * vector-ALU streaming kernel, else generic), 1 = generic only, 2 = no matrix-core kernel.
 * Returns the old value. */
int ep_debug_force_generic_pool(int mode);
/* measurement hook (bench.py `roofline.in_step`): four caller-owned hipEvent_t handles that the calling thread's next
 * ep_head_train_step calls record on their stream in front of / behind the first token pass and in front of / behind the
 * second one (the launches that carry the in-pass contractions and the weight-gradient side work); NULLs switch it off.
 * An event is a marker the queue drains in front of: use it in untimed, instrumented steps only. */
int ep_debug_set_pass_events(void* fwd_begin, void* fwd_end, void* bwd_begin, void* bwd_end);

/* ------------------------------------------------------------------------------------------
 * EP attentive pooling, forward.   Replaces reference poolings/ep.py:35-44 (scores q.k^T,
 * softmax over tokens, attention-weighted reduction), in the pool-then-project form:
 *     S[b,q,n]  = sum_d (cls_token[q,d]*scale) * x[b,n,d]
 *     A[b,q,:]  = softmax_n S[b,q,:]
 *     P[b,q,:]  = sum_n A[b,q,n] * x[b,n,:]                       (B,Q,D)
 * One streaming read of x.  Saved for backward: S (B,Q,N) raw scores and
 * ML (B,Q,4) = {row max m, sum exp(S-m), reserved(delta), 0}.
 * `cls_token` is the learned (Q,D) query block (state_dict key 0.cls_token) or, when
 * cls_bstride != 0, a per-image (B,Q,D) override (the `cls=` argument of ep.py:32-33).
 * `image_index` (optional, NULL = identity): (B,) int32 device array; image b of the batch is
 * x + image_index[b] * x_bstride -- batches are read IN PLACE from a token store resident in HBM
 * (288 GB per GPU), no gather copy.
 */
size_t ep_pool_workspace_bytes(int B, int N, int D, int Q);
/* name of the device kernel ep_pool_forward / ep_pool_backward will launch for this shape (for
 * matching profiler output; static string) */
const char* ep_pool_kernel_name(int B, int N, int D, int Q, int backward);
/* LayerNorm-of-tokens variants.  Several attentive poolers layer-norm every token before their key / value
 * projections (reference poolings/cae_att.py:103-104, poolings/jepa/modules.py:180, poolings/simpool.py:52).  With
 * xhat[b,n] = (x[b,n] - mean) * rstd the passes compute S = (cls*scale) . xhat, P = softmax_n(S) xhat and, backward,
 * dcls = scale * sum dS xhat WITHOUT materialising xhat: per-token {mean, rstd} come from ep_token_stats (one extra
 * streaming read; they depend on the frozen tokens only, so a resident token store computes them once) and enter the
 * passes as two scalars per token.  The LayerNorm's affine part stays outside (it folds into the queries / values).
 * token_stats: (B, N, 2) fp32, or (M, N, 2) indexed by image_index like x. */
int ep_token_stats(const void* x, int x_dtype, int64_t x_bstride, int B, int N, int D, float eps, float* stats,
                   ep_stream_t stream);
int ep_pool_forward_ln(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N,
                       int D, const float* cls_token, int64_t cls_bstride, int Q, float scale,
                       const float* token_stats, float* P, float* S, float* ML, void* workspace,
                       size_t workspace_bytes, ep_stream_t stream);
int ep_pool_backward_ln(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N,
                        int D, int Q, float scale, const float* token_stats, const float* S, const float* ML,
                        const float* dP, float* dcls, int accumulate, void* workspace, size_t workspace_bytes,
                        ep_stream_t stream);
/* same, for a given token storage type (EP_DTYPE_F32 / EP_DTYPE_BF16 / EP_DTYPE_F16) */
const char* ep_pool_kernel_name_ex(int B, int N, int D, int Q, int backward, int x_dtype);
/* name of the device kernel a dense layer y = x W^T (x: M x K, W: N x K, both K-contiguous and 16-byte aligned) launches for
 * this shape: "ep_gemm_b3_kernel" (bf16 x3 at fp32 accuracy, csrc/ep_wgrad3.h: large contractions) or one of the exact-f32
 * kernels (for profiling / the bench's roofline object). */
const char* ep_linear_kernel_name(int M, int N, int K);
int ep_pool_forward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                    int B, int N, int D,
                    const float* cls_token, int64_t cls_bstride, int Q, float scale,
                    float* P, float* S, float* ML,
                    void* workspace, size_t workspace_bytes, ep_stream_t stream);

/* EP pooling, backward w.r.t. cls_token (x is frozen: reference main_linprobe.py:393-400).
 * Replaces autograd of ep.py:39-44.  Second streaming read of x.
 *     dA[b,q,n] = dP[b,q,:] . x[b,n,:] ;  dS = A * (dA - delta[b,q]) ;  delta = ML[b,q,2]
 *     dcls[q,:] (+)= scale * sum_b sum_n dS[b,q,n] * x[b,n,:]
 * accumulate != 0 adds into dcls (gradient accumulation, engine_finetune.py:72-77).      */
int ep_pool_backward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index,
                     int B, int N, int D,
                     int Q, float scale, const float* S, const float* ML, const float* dP,
                     float* dcls, int accumulate,
                     void* workspace, size_t workspace_bytes, ep_stream_t stream);
/* The same second pass for PER-IMAGE query rows (ep_pool_forward with cls_bstride = Q*D; reference poolings/ep.py:32-33,
 * the `cls=` override of EfficientProbing.forward): dq[b,q,:] = scale * sum_n dS[b,q,n] x[b,n,:], one (Q, D) gradient per
 * image, not summed over the batch.  ML[...,2] must hold delta (ep_project_backward writes it).  No workspace. */
int ep_pool_backward_per_image(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N,
                               int D, int Q, float scale, const float* S, const float* ML, const float* dP, float* dq,
                               ep_stream_t stream);

/* Attention maps A = softmax(S) from the saved scores (reference tools/ep_attention_maps.py:52-58). */
int ep_attention_from_scores(const float* S, const float* ML, int B, int Q, int N, float* A,
                             ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-query value projection.  Replaces `self.v(x)` + the slice/concat of ep.py:40,44-45:
 *     y[b, q*Dq + c] = sum_d P[b,q,d] * Wv[q*Dq + c, d]       Dq = Dp / Q, Dp = D / d_out
 * backward:  dP[b,q,d] = sum_c dy[b,q*Dq+c] * Wv[q*Dq+c, d]
 *            dWv[q*Dq+c, d] (+)= sum_b dy[b,q*Dq+c] * P[b,q,d]
 *            ML[b,q,2] = delta[b,q] = sum_c dy[b,q*Dq+c] * y[b,q*Dq+c]                     */
int ep_project_forward(const float* P, const float* Wv, int B, int D, int Dp, int Q, float* y,
                       ep_stream_t stream);
int ep_project_backward(const float* dy, const float* y, const float* P, const float* Wv,
                        int B, int D, int Dp, int Q, float* dP, float* dWv, float* ML,
                        int accumulate, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d(Dp, affine=False, eps) -- reference probe_heads.py:109-110 (torch semantics:
 * biased batch variance for normalisation, unbiased for the running estimate, momentum 0.1).
 * train: z = (y-mu)*rstd ; saves rstd (Dp) ; updates running_mean/var in place and
 *        *num_batches_tracked (int64, device) += 1.
 * backward: dy = rstd * (dz - mean_b dz - z * mean_b(dz*z)).                               */
size_t ep_bn_workspace_bytes(int B, int Dp);
int ep_bn_forward_train(const float* y, int B, int Dp, float eps, float momentum, float* z,
                        float* rstd, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                        ep_stream_t stream);
int ep_bn_forward_eval(const float* y, int B, int Dp, float eps, const float* running_mean,
                       const float* running_var, float* z, ep_stream_t stream);
int ep_bn_backward(const float* dz, const float* z, const float* rstd, int B, int Dp, float* dy,
                   void* workspace, size_t workspace_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Classifier Linear(Dp, C) -- reference probe_heads.py:76.  logits has leading dimension ldl
 * (>= C, multiple of 4; pad columns are written as 0).
 * backward: dz = dlogits Wc ; dWc (+)= dlogits^T z ; dbc (+)= sum_b dlogits.               */
int ep_linear_forward(const float* z, const float* Wc, const float* bc, int B, int Dp, int C,
                      float* logits, int ldl, ep_stream_t stream);
int ep_linear_backward(const float* dlogits, int ldl, const float* z, const float* Wc, int B,
                       int Dp, int C, float* dz, float* dWc, float* dbc, int accumulate,
                       ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pre-split weights ("planes") for the bf16 matrix cores at fp32 accuracy (csrc/ep_planes.hip).
 * For heads with D >= 2048 (or as EP_GEMM_PLANES says) the fused train step splits the head's two
 * weight matrices once per step and runs its contractions (reference poolings/ep.py:40,
 * probe_heads.py:76 and their autograd; the two weight gradients against planes of P^T and z^T)
 * against the planes; these three entry points expose the same kernels.
 *   ep_planes_elems : 16-bit elements of the planes of a rows x K matrix (3 terms, K padded to 32)
 *   ep_planes_split : W (R x K, row-major, leading dimension ldw) -> planes of W (`planes_n`,
 *                     ep_planes_elems(R, K) elements) and / or of W^T (`planes_t`,
 *                     ep_planes_elems(K, R) elements); either may be null; 16-byte aligned
 *   ep_matmul_planes: C (M x N, ldc) = A (M x K fp32, lda % 4 == 0, 16-byte aligned) times the
 *                     TRANSPOSE of the rows_w x K matrix whose planes are given (N <= rows_w),
 *                     + bias[n] when bias != null.  Same result as the fp32 contraction up to
 *                     summation order (every product is exact; terms below 2^-24 are dropped). */
size_t ep_planes_elems(int rows, int K);
int ep_planes_split(const float* W, int R, int K, int64_t ldw, uint16_t* planes_n, uint16_t* planes_t,
                    ep_stream_t stream);
int ep_matmul_planes(const float* A, int64_t lda, const uint16_t* planes, int rows_w, int K,
                     const float* bias, int M, int N, float* C, int64_t ldc, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CrossEntropyLoss() mean reduction + timm accuracy counts -- reference
 * main_linprobe.py:589, engine_finetune.py:62-63.
 *   row_stats[b] = { -log softmax(logits[b])[target[b]] / B, top-1 hit, top-5 hit, non-finite }
 *   (B x 4 floats, caller-owned scratch / output); then, in a fixed order (reproducible),
 *   stats[0] += mean loss, stats[1] += #top-1 hits, stats[2] += #top-5 hits, stats[3] += #rows
 *   with a non-finite logit or loss (the sys.exit(1) test of engine_finetune.py:66-70).
 *   `stats` (4 floats, may be NULL) must be zeroed by the caller when an accounting window starts.
 *   dlogits[b,k] = (softmax - onehot) * grad_scale / B   (grad_scale = loss_scale / accum_iter)
 * dlogits may be NULL (evaluation).                                                        */
int ep_cross_entropy(const float* logits, int ldl, const int64_t* targets, int B, int C,
                     float grad_scale, float* row_stats, float* dlogits, float* stats,
                     ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizers over ONE flat float32 parameter buffer (params, grads, state have the same
 * layout; `segs` describes the tensors inside it).  Replaces reference util/lars.py:13-37 and
 * torch.optim.SGD / AdamW as selected by main_linprobe.py:403-408, plus the GradScaler
 * unscale / inf-skip of util/misc.py:267-277:
 *   g = grad * inv_scale ; if any g is non-finite: *found_inf = 1 and NOTHING is updated.
 *   LARS, for segments with apply_trust (tensor ndim > 1): dp = g + wd*p ;
 *       dp *= tc*||p||/||dp|| if both norms > 0 ; then mu = momentum*mu + dp ; p -= lr*mu.
 * grad_norm_out (optional, 1 float) receives the global L2 norm of the unscaled gradient
 * (util/misc.py:289-301).  found_inf (1 int32, device) is written every call (0 or 1).     */
typedef struct ep_segment {
  int64_t offset;      /* first element of the tensor in the flat buffer (multiple of 4)   */
  int64_t numel;
  int32_t apply_trust; /* 1: weight decay + LARS trust ratio (ndim > 1); 0: plain momentum */
  int32_t reserved;
} ep_segment;

size_t ep_optim_workspace_bytes(int64_t total_numel, int num_segments);
int ep_lars_step(float* params, const float* grads, float* mu, int64_t total_numel,
                 const ep_segment* segs_host, int num_segments, float lr, float weight_decay,
                 float momentum, float trust_coefficient, float inv_scale, int32_t* found_inf,
                 float* grad_norm_out, void* workspace, size_t workspace_bytes,
                 ep_stream_t stream);
int ep_sgd_step(float* params, const float* grads, int64_t total_numel, float lr,
                float weight_decay, float inv_scale, int32_t* found_inf, float* grad_norm_out,
                void* workspace, size_t workspace_bytes, ep_stream_t stream);
int ep_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  int64_t total_numel, int64_t step, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float inv_scale, int32_t* found_inf,
                  float* grad_norm_out, void* workspace, size_t workspace_bytes,
                  ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole train step of Sequential(EP, BN, Linear) + CE (+ optimizer) as one call: the hot loop
 * body of reference engine_finetune.py:52-77.  All buffers caller-owned; `ws` from
 * ep_head_workspace_bytes().  The flat parameter buffer holds, in nn.Module.parameters()
 * order, cls_token (Q*D) | v.weight (Dp*D) | fc.weight (C*Dp) | fc.bias (C); each tensor
 * starts at a multiple of 4 elements (ep_head_param_offsets fills the four offsets and returns
 * the total element count).
 * phases: 1 = forward+loss+backward into `grads` (accumulating if accumulate != 0);
 *         2 = optimizer only; 3 = both.  Between 1 and 2 the caller may all-reduce `grads`
 *         (the single RCCL all-reduce per step that replaces DDP, main_linprobe.py:581-583).
 * Workspace contract (ABI v21): `ws` holds the arrival counters of the in-pass contractions (csrc/ep_inpass.h: the value
 * projection and its dP gradient computed inside the two token passes), which every step leaves at zero again, and the
 * give-up count of their bounded waits.  Before its first use -- and before it serves another (B, N) -- the caller either
 * zero-fills the whole workspace (hipMemset) or calls ep_head_workspace_init (ABI v22: clears exactly those words, on
 * `stream`); it must not write the workspace between steps.  A give-up (nonzero count: a wait on another workgroup of the
 * same launch ran into its bound, e.g. on a partitioned or shared GPU) is LOUD: from then on every optimizer phase run
 * through ep_head_train_step skips its update, sets *found_inf and adds 1 to stats[3] (the non-finite row count that
 * stops the training loop), until the workspace is initialised again. */
typedef struct ep_head_dims {
  int32_t B, N, D, Q, d_out, C;
} ep_head_dims;

typedef struct ep_head_step {
  ep_head_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;   /* optional, see ep_pool_forward */
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;          /* 4 floats, see ep_cross_entropy                                  */
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale;      /* loss_scale / accum_iter applied to dlogits                      */
  float inv_scale;       /* 1 / loss_scale applied by the optimizer                         */
  int32_t accumulate;
  int32_t optimizer;     /* 0 = LARS, 1 = SGD, 2 = AdamW                                    */
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream; /* optional second caller-owned stream: the weight-gradient contractions
                             (dWc, dWv, dbc) that nothing else in the step depends on run there where
                             the second token pass cannot carry them as side workgroups; the call
                             forks and joins with events, `stream` is complete when it returns to
                             the caller's queue order; NULL = everything on `stream`.  The other
                             heads' step structs have the same field: since round 4 their
                             parameter-gradient contractions start there as soon as their operands
                             exist (csrc/ep_internal.h: AuxSide). */
  /* Split phases for communication overlap (ABI v7).  phases bit 2 (4): only the first token pass
   * (needs nothing but cls_token); bit 3 (8): everything of phase 1 after it.  1 == 4|8.
   * The optimizer phase updates parameter tensors [opt_first_segment, opt_first_segment +
   * opt_num_segments) of the flat layout (0, 0 = all four): a data-parallel caller all-reduces the
   * cls_token gradient first, updates it, starts the next step's first token pass and lets the large
   * all-reduce of the remaining gradients run beside it (see engine.ProbeHeadEngine).  found_inf /
   * grad_norm then cover the updated tensors only. */
  int32_t opt_first_segment, opt_num_segments;
  /* Deferred large update (ABI v21).  phases bit 4 (16), with phases bit 1 (2) and an aux_stream: the optimizer updates
   * cls_token on `stream` (one launch), then v.weight / fc.weight / fc.bias on aux_stream, and records the caller-owned
   * event `defer_event` (a hipEvent_t) there instead of joining -- the next step's first token pass needs nothing but
   * cls_token, so it starts right behind the small update while the large one runs beside it.  phases bit 5 (32) on a
   * LATER call: make `stream` wait for defer_event in front of the first read of those three tensors (behind the first
   * token pass).  A caller that reads the parameters itself waits on the event first (engine.ProbeHeadEngine.flush()).
   * Same arithmetic and order per tensor as the undeferred step; found_inf / grad_norm then cover the tensors of the call
   * that wrote them last (use it without loss scaling only). */
  void* defer_event;
  /* bf16 planes of the weight matrices (csrc/ep_planes.hip: the operands of the bf16 x3 contractions, used for D >= 2048 and
   * by EP_GEMM_PLANES) live in `ws`.  Every optimizer phase run through this call WRITES the planes of the tensors it updates
   * (ABI v22: the update kernel emits them tile by tile, no split launch).  planes_valid != 0 tells a forward/backward phase
   * that the planes in `ws` are those of the current parameters -- i.e. the last writer of v.weight / fc.weight was an
   * optimizer phase of this call on this workspace (or the parameters are unchanged since a forward phase established
   * them); 0 (the safe default) makes the step split them itself, as before.  engine.ProbeHeadEngine tracks it through the
   * parameters' torch version counter. */
  int32_t planes_valid;
  /* ABI v25: arithmetic of the step's six contractions (y = P Wv^T, logits, dz, dP, dWv, dWc -- reference poolings/ep.py:40,
   * probe_heads.py:76 and their autograd).  EP_ARITH_F32 (0): fp32 results (exact-fp32 or bf16 x3 matrix-core products).
   * EP_ARITH_BF16_AUTOCAST (1): what the published runs' --amp bfloat16 does inside autocast (reference
   * engine_finetune.py:52-55): both operands rounded to bf16, ONE matrix-core product, fp32 accumulation -- the outputs stay
   * fp32 (autocast rounds them to bf16 as well), softmax / BatchNorm statistics / cross-entropy / the optimizer stay fp32, the
   * token passes keep their arithmetic.  Every slice width since round 6: where (D / d_out / Q) % 32 != 0 (256 x 768 at the
   * protocol's 32 queries, SigLIP2 SO400M's 1152) dP = dy_q Wv_q -- the one contraction over a query's SLICE of the planes'
   * permuted k-order -- runs on the thin-slice / single-product tile kernels instead.  A secondary mode: never the default,
   * never the headline number. */
  int32_t arith;
  /* ABI v26: device-resident loss scale -- torch.cuda.amp.GradScaler (reference util/misc.py:260-286: scale(loss), unscale_,
   * step skipped on inf / nan, update()) without a host read per step.  scaler_state (device, 4 floats, NULL = off): two slots
   * of {scale, growth tracker}.  A step READS slot scaler_slot: the loss gradient is scaled by grad_scale * scale, the
   * optimizer unscales by inv_scale / scale (grad_scale / inv_scale then carry only 1 / accum_iter and 1 / world); its
   * optimizer phase WRITES the other slot: scale * scaler_backoff and tracker 0 when a gradient was non-finite (the update is
   * skipped, *found_inf = 1), else tracker + 1, and scale * scaler_growth with tracker 0 once it reaches scaler_interval.  The
   * caller flips scaler_slot after every optimizer phase and reads the state back when it needs it (checkpoints).  Whole-tensor
   * optimizer phases only (opt_num_segments == 0, no deferred update): EP_E_UNSUPPORTED otherwise. */
  float* scaler_state;
  int32_t scaler_slot;
  float scaler_growth, scaler_backoff;
  int32_t scaler_interval;
} ep_head_step;

int64_t ep_head_param_offsets(const ep_head_dims* dims, int64_t offsets[4]);
size_t ep_head_workspace_bytes(const ep_head_dims* dims);
/* byte offset inside `ws` of an int32 that counts bounded flag waits of the in-pass contractions that gave up (always 0
 * in a correct run; a diagnostic for tests).  -1: bad dims. */
int64_t ep_head_workspace_flag_offset(const ep_head_dims* dims);
/* byte offset inside `ws` of the train-mode logits (B x ldl floats, *ldl = row stride) the last forward phase left there --
 * what the loss of that step was computed from (a read-only view for tests and diagnostics; ABI v25).  -1: bad dims. */
int64_t ep_head_workspace_logits_offset(const ep_head_dims* dims, int32_t* ldl);
/* clear the counters the step keeps in `ws` (see "Workspace contract" above); asynchronous on `stream` */
int ep_head_workspace_init(const ep_head_dims* dims, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_head_train_step(const ep_head_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
/* eval forward: logits (B, ldl) from tokens using running statistics
 * (reference engine_finetune.py:106-166 inner forward).                                    */
int ep_head_eval_forward(const ep_head_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* params, const float* running_mean,
                         const float* running_var, float bn_eps, float* logits, int ldl,
                         void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CoCa attentional pooler head (reference poolings/coca_pytorch.py:250-343 CrossAttention with the
 * registry's arguments probe_heads.py:78: dim_head 64, heads 8, 196 image queries, no context norm,
 * no parallel feed-forward; `return out[:, 0]`).  Only query 0 reaches the output, the keys / values
 * are one shared head, so the module is algebraically the EP token pass with H derived query rows
 *     u[h] = scale * Wk^T (to_q(LayerNorm(img_queries[0])))[h]            (batch independent)
 *     P[b,h] = softmax_n(u[h] . x[b,n]) x[b]      o[b,h] = P[b,h] Wv^T      y[b] = to_out(concat_h o[b,h])
 * and runs on the same streaming kernels: two passes over the tokens per train step.
 *
 * Pooler parameters are five separate tensors:
 *   gamma (D) | img_queries (M,D) | to_q.weight (H*dh, D) | to_kv.weight (2*dh, D) | to_out.weight (D, H*dh)
 * `beta` is the LayerNorm buffer (zeros in the reference; NULL = zeros).  Rows 1..M-1 of img_queries
 * receive zero gradient (they are still part of the LARS norm of the tensor).
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_coca_dims {
  int32_t B, N, D, H, dh, M, C;      /* C used only by the whole-head entry points */
} ep_coca_dims;

typedef struct ep_coca_params {
  float* gamma; const float* beta; float* img_queries; float* to_q; float* to_kv; float* to_out;
} ep_coca_params;

size_t ep_coca_pool_workspace_bytes(const ep_coca_dims* dims);
/* y (B, D) = pooler(x).  `ws` keeps what the backward needs (same ws must be passed to it). */
int ep_coca_pool_forward(const ep_coca_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const ep_coca_params* params, float ln_eps,
                         float* y, void* ws, size_t ws_bytes, ep_stream_t stream);
/* grads (same five-tensor layout; beta ignored) (+)= d loss / d params for upstream dy (B, D). */
int ep_coca_pool_backward(const ep_coca_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const int32_t* image_index, const ep_coca_params* params, const float* dy,
                          const ep_coca_params* grads, int accumulate, void* ws, size_t ws_bytes,
                          ep_stream_t stream);
/* attention of query 0: A (B, H, N) = softmax over tokens (for attention maps / tests) */
int ep_coca_attention(const ep_coca_dims* dims, const void* ws, float* A, ep_stream_t stream);

/* Whole train step of Sequential(CrossAttention, BN, Linear) + CE (+ optimizer), the counterpart of
 * ep_head_train_step.  Flat parameter buffer, each tensor starting at a multiple of 4 elements:
 *   gamma | img_queries | to_q.weight | to_kv.weight | to_out.weight | fc.weight (C,D) | fc.bias (C)
 * (ep_coca_head_param_offsets fills the seven offsets and returns the total element count).        */
typedef struct ep_coca_step {
  ep_coca_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  const float* ln_beta; float ln_eps;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
  int32_t arith;         /* ABI v26: EP_ARITH_F32 / EP_ARITH_BF16_AUTOCAST, as ep_head_step.arith -- the step's contractions (to_out,
                            the per-head value projection, the classifier and their gradients) as ONE bf16 product with fp32
                            accumulation; the token passes, LayerNorm, softmax, BatchNorm, loss, optimizer stay fp32 */
} ep_coca_step;

int64_t ep_coca_head_param_offsets(const ep_coca_dims* dims, int64_t offsets[7]);
size_t ep_coca_head_workspace_bytes(const ep_coca_dims* dims);
/* as ep_head_workspace_logits_offset: where the last forward phase left its train-mode logits inside `ws` (ABI v26) */
int64_t ep_coca_head_workspace_logits_offset(const ep_coca_dims* dims, int32_t* ldl);
int ep_coca_head_train_step(const ep_coca_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_coca_head_eval_forward(const ep_coca_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                              const int32_t* image_index, const float* params, const float* ln_beta,
                              float ln_eps, const float* running_mean, const float* running_var,
                              float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                              ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * AbMILP head (reference poolings/abmilp.py:11-75 with models_vit.Attention :43-97, num_heads = 1, no
 * qkv bias; registry arguments probe_heads.py:42-51 at their defaults: self-attention applied to
 * "both", tanh predictor of depth 2, no positional conditioning):
 *     Xa = proj(softmax((x Wq^T * D^-1/2)(x Wk^T)^T)(x Wv^T)) + b_p                     (B,N,D)
 *     a  = softmax_n(w2 . tanh(W1 Xa + b1) + b2)                                          (B,N)
 *     out[b] = sum_n a[b,n] Xa[b,n,:]                                                     (B,D)
 * Dense and matrix-core bound (about 3.7 GFLOP per image forward at N = 256, D = 1152): every
 * contraction runs on the exact-fp32 MFMA kernel of this library, per image for the N x N attention
 * and over all B*N token rows for the projections and their weight gradients.
 * Tokens must be contiguous (x_bstride == N*D).
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_abmilp_dims {
  int32_t B, N, D, C;                 /* C used only by the whole-head entry points */
} ep_abmilp_dims;

typedef struct ep_abmilp_params {
  float* qkv;      /* self_attn.qkv.weight (3D, D)            */
  float* proj_w;   /* self_attn.proj.weight (D, D)            */
  float* proj_b;   /* self_attn.proj.bias (D)                 */
  float* w1;       /* attention_predictor.0.weight (D, D)     */
  float* b1;       /* attention_predictor.0.bias (D)          */
  float* w2;       /* attention_predictor.2.weight (1, D)     */
  float* b2;       /* attention_predictor.2.bias (1)          */
} ep_abmilp_params;

size_t ep_abmilp_pool_workspace_bytes(const ep_abmilp_dims* dims);
/* out (B, D); attn_map (B, N) optional (NULL to skip): the predictor's softmax weights
 * (abmilp.py:56-66 forward_with_attn_map).  `ws` keeps what the backward needs. */
int ep_abmilp_pool_forward(const ep_abmilp_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                           const ep_abmilp_params* params, float* out, float* attn_map, void* ws,
                           size_t ws_bytes, ep_stream_t stream);
int ep_abmilp_pool_backward(const ep_abmilp_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const ep_abmilp_params* params, const float* dout,
                            const ep_abmilp_params* grads, int accumulate, void* ws, size_t ws_bytes,
                            ep_stream_t stream);

/* Whole train step of Sequential(ABMILPHead, BN, Linear) + CE (+ optimizer).  Flat parameter buffer:
 *   qkv | proj_w | proj_b | w1 | b1 | w2 | b2 | fc.weight (C,D) | fc.bias (C)      (nine offsets) */
typedef struct ep_abmilp_step {
  ep_abmilp_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  int32_t arith;         /* ABI v26: EP_ARITH_F32 / EP_ARITH_BF16_AUTOCAST -- every contraction of the step (qkv, q k^T, A v, proj,
                            predictor, classifier and all their gradients) as ONE bf16 matrix-core product with fp32 accumulation
                            (reference engine_finetune.py:52-55: the published runs' --amp bfloat16); softmax, tanh, pooling,
                            BatchNorm, loss and optimizer stay fp32, outputs are not rounded */
} ep_abmilp_step;

int64_t ep_abmilp_head_param_offsets(const ep_abmilp_dims* dims, int64_t offsets[9]);
size_t ep_abmilp_head_workspace_bytes(const ep_abmilp_dims* dims);
int64_t ep_abmilp_head_workspace_logits_offset(const ep_abmilp_dims* dims, int32_t* ldl);   /* (ABI v26, as for the EP head) */
int ep_abmilp_head_train_step(const ep_abmilp_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_abmilp_head_eval_forward(const ep_abmilp_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                                const float* params, const float* running_mean, const float* running_var,
                                float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                                ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weighted k-NN classifier on frozen features (reference engine_finetune.py:224-266 knn_classifier;
 * sweep over k and T in main_linprobe.py:411-465).  Features are row-major fp32, normally L2-normalised
 * (ep_l2_normalize = torch.nn.functional.normalize, main_linprobe.py:441-442).
 *   ep_knn_topk : sims (M, out_ld), idx (M, out_ld): the k largest test.train^T similarities of every
 *                 test row, sorted descending (ties: lower train index first) -- `similarity.topk(k)`.
 *                 Exact fp32; the similarity matrix is produced in row chunks inside `ws`.
 *   ep_knn_vote : for a prefix k of those lists: probs[c] = sum exp(sim/T) over neighbours of label c,
 *                 pred (M, 5) = the five best classes; with `targets`, counts[0] += #top-1 hits and
 *                 counts[1] += #top-5 hits (float counters).
 * ------------------------------------------------------------------------------------------ */
int ep_l2_normalize(const float* x, int64_t rows, int D, float eps, float* out, ep_stream_t stream);
size_t ep_knn_workspace_bytes(int M, int n_train);
/* the same for features of width D (round 4: the gallery's bf16 planes live in the workspace too -- they are split once per
 * search and every query chunk's similarities run on the planes kernel; ep_knn_workspace_bytes sizes them for D <= 1536) */
size_t ep_knn_workspace_bytes_ex(int M, int n_train, int D);
int ep_knn_topk(const float* test, const float* train, int M, int n_train, int D, int k, float* sims,
                int32_t* idx, int out_ld, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_knn_vote(const float* sims, const int32_t* idx, int ld, const int64_t* train_labels, int M, int k,
                float T, int num_classes, const int64_t* targets, int32_t* pred, float* counts,
                ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Plain linear probing: Sequential(BatchNorm1d(affine=False, eps=1e-6), Linear) on one feature vector
 * per image -- what reference probe_heads.py:96-99 builds for --cls_features cls / gap / pos / ... (no
 * attentive pooling).  Flat parameter buffer: fc.weight (C, D) | fc.bias (C).  Same step struct as the EP
 * head (dims.N, dims.Q, dims.d_out are ignored; x is the (B, D) fp32 feature matrix, x_bstride = D).
 * ------------------------------------------------------------------------------------------ */
int64_t ep_lp_param_offsets(const ep_head_dims* dims, int64_t offsets[2]);
size_t ep_lp_workspace_bytes(const ep_head_dims* dims);
int ep_lp_train_step(const ep_head_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_lp_eval_forward(const ep_head_dims* dims, const float* x, const float* params, const float* running_mean,
                       const float* running_var, float bn_eps, float* logits, int ldl, void* ws,
                       size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SigLIP attention-pool head (reference poolings/clip/attention_pool.py:13-140 AttentionPoolLatent as the
 * registry builds it, probe_heads.py:72: AttentionPoolLatent(in_features=dim) -> 8 heads, one latent query,
 * qkv bias, mlp_ratio 4, no norms, pool 'token').  With u_h = scale Wk_h^T q_h (q = q(latent)):
 *     o[b,h] = softmax_n(u_h . x[b,n]) x[b] Wv_h^T + bv_h      z1 = proj(o)      out = z1 + mlp(z1)
 * -- the EP token passes with H derived query rows and the EP per-query value projection, then two small
 * dense layers per image.  Eleven pooler tensors (nn.Module state-dict names):
 *   latent (1,1,D) | q.weight q.bias | kv.weight (2D,D) kv.bias (2D) | proj.weight proj.bias |
 *   mlp.fc1.weight (hidden,D) mlp.fc1.bias | mlp.fc2.weight (D,hidden) mlp.fc2.bias
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_siglip_dims {
  int32_t B, N, D, H, hidden, C;     /* C used only by the whole-head entry points */
} ep_siglip_dims;

typedef struct ep_siglip_params {
  float *latent, *q_w, *q_b, *kv_w, *kv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} ep_siglip_params;

size_t ep_siglip_pool_workspace_bytes(const ep_siglip_dims* dims);
int ep_siglip_pool_forward(const ep_siglip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                           const int32_t* image_index, const ep_siglip_params* params, float* out, void* ws,
                           size_t ws_bytes, ep_stream_t stream);
int ep_siglip_pool_backward(const ep_siglip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const int32_t* image_index, const ep_siglip_params* params, const float* dout,
                            const ep_siglip_params* grads, int accumulate, void* ws, size_t ws_bytes,
                            ep_stream_t stream);
/* attention of the latent query: A (B, H, N) */
int ep_siglip_attention(const ep_siglip_dims* dims, const void* ws, float* A, ep_stream_t stream);

/* Whole train step of Sequential(AttentionPoolLatent, BN, Linear) + CE (+ optimizer).  Flat parameter buffer:
 * the eleven pooler tensors in the order above | fc.weight (C,D) | fc.bias (C)   (thirteen offsets).
 * Same fields as ep_coca_step without the LayerNorm ones. */
typedef struct ep_siglip_step {
  ep_siglip_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_siglip_step;

int64_t ep_siglip_head_param_offsets(const ep_siglip_dims* dims, int64_t offsets[13]);
size_t ep_siglip_head_workspace_bytes(const ep_siglip_dims* dims);
int ep_siglip_head_train_step(const ep_siglip_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_siglip_head_eval_forward(const ep_siglip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                                const int32_t* image_index, const float* params, const float* running_mean,
                                const float* running_var, float bn_eps, float* logits, int ldl, void* ws,
                                size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CAE attentive block (reference poolings/cae_att.py:79-108 CAEAttentiveBlock with CrossAttention :19-77 as the
 * registry builds it, probe_heads.py:83: CAEAttentiveBlock(dim=dim) -> 8 heads, no qkv bias).  One learned query
 * token; keys LN_k(x) Wk^T, values LN_v(x) Wv^T: two LayerNorms of the same token, i.e. one normalised token xhat with
 * two affine maps -- the LayerNorm-of-tokens mode of the EP passes (ep_pool_forward_ln) with derived query rows
 * gk * (Wk_h^T qh_h), then the per-head projection with Wv diag(gv) and bias Wv bv, then proj.
 * Fourteen tensors: query_token (1,1,D) | norm1_q.weight .bias | norm1_k.weight .bias | norm1_v.weight .bias |
 * norm2_cross.weight .bias (unused by the forward: zero gradient) | cross_attn.q.weight k.weight v.weight (D,D) |
 * cross_attn.proj.weight (D,D) .bias.  token_stats: optional (B|M, N, 2) from ep_token_stats (NULL: computed here).
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_cae_dims {
  int32_t B, N, D, H, C;
} ep_cae_dims;

typedef struct ep_cae_params {
  float *query, *nq_w, *nq_b, *nk_w, *nk_b, *nv_w, *nv_b, *n2_w, *n2_b, *q_w, *k_w, *v_w, *proj_w, *proj_b;
} ep_cae_params;

size_t ep_cae_pool_workspace_bytes(const ep_cae_dims* dims);
int ep_cae_pool_forward(const ep_cae_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                        const int32_t* image_index, const float* token_stats, float ln_eps,
                        const ep_cae_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_cae_pool_backward(const ep_cae_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* token_stats, float ln_eps,
                         const ep_cae_params* params, const float* dy, const ep_cae_params* grads, int accumulate,
                         void* ws, size_t ws_bytes, ep_stream_t stream);

typedef struct ep_cae_step {
  ep_cae_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* token_stats; float ln_eps;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_cae_step;

int64_t ep_cae_head_param_offsets(const ep_cae_dims* dims, int64_t offsets[16]);
size_t ep_cae_head_workspace_bytes(const ep_cae_dims* dims);
int ep_cae_head_train_step(const ep_cae_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_cae_head_eval_forward(const ep_cae_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                             const int32_t* image_index, const float* token_stats, float ln_eps, const float* params,
                             const float* running_mean, const float* running_var, float bn_eps, float* logits,
                             int ldl, void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * V-JEPA attentive pooler (reference poolings/jepa/attentive_pooler.py:21-104 with CrossAttentionBlock / CrossAttention /
 * MLP of poolings/jepa/modules.py:13-183 as the registry builds it, probe_heads.py:81: AttentivePooler(embed_dim=dim,
 * num_heads=args.num_heads): one query token, depth 1, complete block, qkv bias):
 *     y = xattn(q0, LN1(x));  q1 = q0 + y;  out = q1 + mlp(LN2(q1))
 * = the SigLIP head with LayerNorm-ed keys / values (LayerNorm-of-tokens mode of the passes), the residual with the
 * query token and a LayerNorm in front of the MLP.  Fifteen tensors: query_tokens (1,1,D) | norm1.weight .bias |
 * xattn.q.weight .bias | xattn.kv.weight (2D,D) .bias | xattn.proj.weight .bias | norm2.weight .bias |
 * mlp.fc1.weight (hidden,D) .bias | mlp.fc2.weight (D,hidden) .bias.
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_jepa_dims {
  int32_t B, N, D, H, hidden, C;
} ep_jepa_dims;

typedef struct ep_jepa_params {
  float *query, *n1_w, *n1_b, *q_w, *q_b, *kv_w, *kv_b, *proj_w, *proj_b, *n2_w, *n2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} ep_jepa_params;

size_t ep_jepa_pool_workspace_bytes(const ep_jepa_dims* dims);
int ep_jepa_pool_forward(const ep_jepa_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* token_stats, float ln_eps,
                         const ep_jepa_params* params, float* out, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_jepa_pool_backward(const ep_jepa_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const int32_t* image_index, const float* token_stats, const ep_jepa_params* params,
                          const float* dout, const ep_jepa_params* grads, int accumulate, void* ws, size_t ws_bytes,
                          ep_stream_t stream);

typedef struct ep_jepa_step {
  ep_jepa_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* token_stats; float ln_eps;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_jepa_step;

int64_t ep_jepa_head_param_offsets(const ep_jepa_dims* dims, int64_t offsets[17]);
size_t ep_jepa_head_workspace_bytes(const ep_jepa_dims* dims);
int ep_jepa_head_train_step(const ep_jepa_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_jepa_head_eval_forward(const ep_jepa_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                              const int32_t* image_index, const float* token_stats, float ln_eps, const float* params,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits,
                              int ldl, void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * AIM attention-pooling head (reference poolings/aim.py:337-392 AttentionPoolingClassifier as the registry builds it,
 * probe_heads.py:73: AttentionPoolingClassifier(dim=dim, num_heads=args.num_heads); one learned query token split into
 * H heads, keys / values = Linear(dim, dim, bias=False) of the per-channel batch-normalised tokens, aim.py:364-391).
 * The token BatchNorm1d(dim, affine=False, eps=1e-6) has no parameters and the tokens are frozen, so its statistics
 * mu, r are constants of the step: the head is the plain EP token pass with derived query rows r * (scale Wk_h^T q_h),
 * followed by the per-head projection with Wv diag(r) and bias -Wv (mu r).  Three tensors: cls_token (1,1,D) |
 * k.weight (D,D) | v.weight (D,D); the BatchNorm's running_mean / running_var / num_batches_tracked are buffers.
 *
 * ep_channel_stats: per-image column statistics image_stats[b] = {mean_n x[b,n,:], sum_n (x[b,n,:] - mean)^2}
 * ((B, 2, D) floats).  They depend on the frozen tokens only: a resident token store computes them ONCE for all its
 * images and passes the (M, 2, D) table as `image_stats` (rows are then addressed through image_index like the
 * tokens); with image_stats == NULL the step computes them for the batch (one more streaming read of the tokens).
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_aim_dims {
  int32_t B, N, D, H, C;
} ep_aim_dims;

typedef struct ep_aim_params {
  float *cls_token, *k_w, *v_w;
} ep_aim_params;

int ep_channel_stats(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                     float* image_stats, ep_stream_t stream);
size_t ep_aim_pool_workspace_bytes(const ep_aim_dims* dims);
/* training != 0: batch statistics (running_* updated with `bn_momentum`, unbiased variance, num_batches_tracked += 1;
 * running pointers may be NULL); training == 0: the running statistics normalise. */
int ep_aim_pool_forward(const ep_aim_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                        const int32_t* image_index, const float* image_stats, int training, float bn_eps,
                        float bn_momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        const ep_aim_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream);
/* `y`: the output of the matching ep_aim_pool_forward call on the same workspace */
int ep_aim_pool_backward(const ep_aim_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const ep_aim_params* params, const float* y, const float* dy,
                         const ep_aim_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream);
/* attention weights (B, H, N) of the last forward on this workspace (tools/ep_attention_maps.py counterpart) */
int ep_aim_attention(const ep_aim_dims* dims, const void* ws, float* A, ep_stream_t stream);

typedef struct ep_aim_step {
  ep_aim_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* image_stats;                          /* optional cached (M, 2, D) table of ep_channel_stats */
  float* tok_running_mean; float* tok_running_var; int64_t* tok_num_batches_tracked;   /* 0.bn.* buffers */
  float tok_bn_eps, tok_bn_momentum;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_aim_step;

/* flat layout: cls_token | k.weight | v.weight | fc.weight | fc.bias */
int64_t ep_aim_head_param_offsets(const ep_aim_dims* dims, int64_t offsets[5]);
size_t ep_aim_head_workspace_bytes(const ep_aim_dims* dims);
int ep_aim_head_train_step(const ep_aim_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_aim_head_eval_forward(const ep_aim_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                             const int32_t* image_index, float tok_bn_eps, const float* tok_running_mean,
                             const float* tok_running_var, const float* params, const float* running_mean,
                             const float* running_var, float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes,
                             ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SimPool heads (reference poolings/simpool.py; registry entries probe_heads.py:66-70):
 *   linears = 1: SimPool(dim, num_heads=1, qkv_bias=False, gamma=None)        (simpool.py:5-91,  --cls_features simpool)
 *   linears = 0: SimPool_nolinears(dim, num_heads=12, gamma=None)             (simpool.py:93-170, --cls_features esimpool)
 * The query is derived from the image's own mean token, so the query rows differ per image: these heads run on the
 * per-image-query token passes (csrc/ep_pool_imgq.hip) with LayerNorm-of-tokens scores (eps 1e-6); simpool pools the
 * normalised tokens (then applies the LayerNorm's affine part), esimpool pools the raw tokens per head slice.
 * Tensors: norm_patches.weight .bias (D) [| wq.weight wk.weight (D,D) when linears].
 * token_stats ((M|B, N, 2) of ep_token_stats with eps = ln_eps) and image_stats ((M|B, 2, D) of ep_channel_stats; row 0 =
 * the mean token) are optional per-store tables addressed through image_index like the tokens; NULL: computed for the
 * batch (token_stats must be given when image_index is).  `y` of the backward = the matching forward's output.
 * ------------------------------------------------------------------------------------------ */
/* The token passes underneath, usable on their own: per-image query rows u (B, D) over H channel slices (head h scores and
 * pools channels [h D/H, (h+1) D/H) only; every token is read once for all heads).  token_stats != NULL: scores on the
 * normalised tokens; pool_ln: pool the normalised tokens (else the raw ones).  Outputs P (B, D) pooled slices and
 * ML (B, H, 2) = {running max, sum of exponentials}; the backward takes dP (B, D) and returns du (B, D) PER IMAGE. */
int ep_imgq_pool_forward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                         int H, const float* u, const float* token_stats, int pool_ln, float* P, float* ML,
                         ep_stream_t stream);
int ep_imgq_pool_backward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                          int H, const float* u, const float* token_stats, int pool_ln, const float* P, const float* ML,
                          const float* dP, float* du, ep_stream_t stream);

/* FULL-WIDTH per-image query rows (the CLIP head's passes, usable on their own): Q <= 4 rows u (B, Q, D) per image, each
 * scoring and pooling the whole token row; every token is read once for all rows (csrc/ep_pool_imgq.hip: ep_imgqf_kernel).
 *   forward : S[b,q,n] = u[b,q] . k[b,n] (+ score_bias[b,q,n]) ; A = softmax_n S ; P[b,q] = sum_n A k[b,n]
 *             k = x, or the normalised token xhat when token_stats != NULL.  Outputs P (B,Q,D), S (B,Q,N) and
 *             ML (B,Q,4) = {row max, sum of exponentials, 0, 0}.
 *   backward: dA = dP[b,q] . k[b,n] (+ dA_bias[b,q,n]) ; dS = A (dA - ML[b,q,2]) with A from the stored S and ML[b,q,0:2]
 *             (the caller puts its softmax correction term delta into ML[...,2]); du[b,q] = sum_n dS k[b,n] PER IMAGE;
 *             dS_out (B,Q,N) optional.
 * D % 4 == 0, D <= 1280. */
int ep_rowq_pool_forward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                         int Q, const float* u, const float* token_stats, const float* score_bias, float* P, float* S,
                         float* ML, ep_stream_t stream);
int ep_rowq_pool_backward(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                          int Q, const float* token_stats, const float* S, const float* ML, const float* dP,
                          const float* dA_bias, float* dS_out, float* du, ep_stream_t stream);

typedef struct ep_simpool_dims {
  int32_t B, N, D, H, C, linears;
} ep_simpool_dims;

typedef struct ep_simpool_params {
  float *norm_w, *norm_b, *wq, *wk;
} ep_simpool_params;

size_t ep_simpool_pool_workspace_bytes(const ep_simpool_dims* dims);
int ep_simpool_pool_forward(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const int32_t* image_index, const float* token_stats, const float* image_stats, float ln_eps,
                            const ep_simpool_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_simpool_pool_backward(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                             const int32_t* image_index, const float* token_stats, const float* image_stats, float ln_eps,
                             const ep_simpool_params* params, const float* y, const float* dy,
                             const ep_simpool_params* grads, int accumulate, void* ws, size_t ws_bytes,
                             ep_stream_t stream);
/* attention weights (B, H, N) of the last forward on this workspace (simpool.py return_attn) */
int ep_simpool_attention(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* token_stats, const void* ws, float* A,
                         ep_stream_t stream);

typedef struct ep_simpool_step {
  ep_simpool_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* token_stats; const float* image_stats; float ln_eps;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_simpool_step;

/* flat layout: norm_patches.weight | .bias | wq.weight | wk.weight (empty without linears) | fc.weight | fc.bias */
int64_t ep_simpool_head_param_offsets(const ep_simpool_dims* dims, int64_t offsets[6]);
size_t ep_simpool_head_workspace_bytes(const ep_simpool_dims* dims);
int ep_simpool_head_train_step(const ep_simpool_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_simpool_head_eval_forward(const ep_simpool_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                                 const int32_t* image_index, const float* token_stats, const float* image_stats,
                                 float ln_eps, const float* params, const float* running_mean, const float* running_var,
                                 float bn_eps, float* logits, int ldl, void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CaiT class-attention pooling (reference poolings/other_pool.py:390-507 CAPooling with one LayerScale_Block_CA /
 * Class_Attention as the registry builds it, probe_heads.py:79: CAPooling(embed_dim=dim) -> 4 heads, qkv bias, LayerNorm
 * eps 1e-6 inside the block, LayerScale vectors, MLP x4 (hidden), final LayerNorm eps 1e-5):
 *     u = norm1([cls ; x]) ; a = proj(softmax(scale q(u_0) k(u)^T) v(u)) ; c1 = cls + gamma_1 a ;
 *     c2 = c1 + gamma_2 mlp(norm2(c1)) ; out = norm(c2)
 * = the LayerNorm-of-tokens mode of the EP passes with derived query rows, plus ONE extra softmax entry per head (the
 * class row itself, batch independent) merged into the pass's softmax state.  Twenty-one tensors: cls_token (1,1,D) |
 * gamma_1 gamma_2 | norm1.weight .bias | attn.q.weight .bias attn.k.weight .bias attn.v.weight .bias | attn.proj.weight
 * .bias | norm2.weight .bias | mlp.fc1.weight (hidden,D) .bias mlp.fc2.weight (D,hidden) .bias | norm.weight .bias.
 * token_stats: optional (B|M, N, 2) from ep_token_stats with eps = ln_eps.
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_cait_dims {
  int32_t B, N, D, H, hidden, C;
  float ln_eps, final_eps;
} ep_cait_dims;

typedef struct ep_cait_params {
  float *cls_token, *gamma_1, *gamma_2, *n1_w, *n1_b, *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *proj_w, *proj_b, *n2_w, *n2_b,
      *fc1_w, *fc1_b, *fc2_w, *fc2_b, *norm_w, *norm_b;
} ep_cait_params;

size_t ep_cait_pool_workspace_bytes(const ep_cait_dims* dims);
int ep_cait_pool_forward(const ep_cait_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* token_stats, const ep_cait_params* params, float* out,
                         void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_cait_pool_backward(const ep_cait_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const int32_t* image_index, const float* token_stats, const ep_cait_params* params,
                          const float* dout, const ep_cait_params* grads, int accumulate, void* ws, size_t ws_bytes,
                          ep_stream_t stream);

typedef struct ep_cait_step {
  ep_cait_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* token_stats; float ln_eps;            /* ln_eps: unused (dims.ln_eps rules); kept for layout parity */
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_cait_step;

int64_t ep_cait_head_param_offsets(const ep_cait_dims* dims, int64_t offsets[23]);
size_t ep_cait_head_workspace_bytes(const ep_cait_dims* dims);
int ep_cait_head_train_step(const ep_cait_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_cait_head_eval_forward(const ep_cait_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                              const int32_t* image_index, const float* token_stats, const float* params,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                              void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CLIP attention pooling (reference poolings/clip/attention_pool2d.py:100-169 AttentionPool2d as the registry builds it,
 * probe_heads.py:54-57,71: AttentionPool2d(in_features=dim, feat_size=14 | 16) -> 4 heads, qkv bias, learned absolute
 * position embedding (N + 1, D) with N = feat_size^2, LayerNorm eps 1e-6 in front).  Rows: t_0 = mean_n LN(x)_n + pos_0,
 * t_n = LN(x)_n + pos_n; only the attention output of row 0 is returned.  Per-image full-width query rows
 * u[b,h] = g * (scale Wk_h^T q0[b]_h) on the normalised tokens + additive score bias w . pos_n + the mean row as one extra
 * softmax entry + the position-embedding part of the values as A . pos (csrc/ep_clip.hip).
 * Seven tensors: pos_embed (N+1, D) | qkv.weight (3D, D) .bias (3D) | proj.weight (D, D) .bias | norm.weight .bias.
 * token_stats: optional (B|M, N, 2) from ep_token_stats with eps = ln_eps.
 * ep_token_xhat_mean: xbar[b] = mean_n (x[b,n] - mean_n) rstd_n, the per-image mean of the normalised tokens.
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_clip_dims {
  int32_t B, N, D, H, C;
  float ln_eps;
} ep_clip_dims;

typedef struct ep_clip_params {
  float *pos_embed, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm_w, *norm_b;
} ep_clip_params;

int ep_token_xhat_mean(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, const float* token_stats,
                       int B, int N, int D, float* xbar, ep_stream_t stream);
size_t ep_clip_pool_workspace_bytes(const ep_clip_dims* dims);
int ep_clip_pool_forward(const ep_clip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* token_stats, const ep_clip_params* params, float* y,
                         void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_clip_pool_backward(const ep_clip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const int32_t* image_index, const float* token_stats, const ep_clip_params* params,
                          const float* dy, const ep_clip_params* grads, int accumulate, void* ws, size_t ws_bytes,
                          ep_stream_t stream);
/* attention of the mean-row query over the patch rows (B, H, N) of the last forward on this workspace (return_attn) */
int ep_clip_attention(const ep_clip_dims* dims, const void* ws, float* A, ep_stream_t stream);

typedef struct ep_clip_step {
  ep_clip_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* token_stats; float ln_eps;            /* ln_eps: unused (dims.ln_eps rules); kept for layout parity */
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
  /* ABI v23: optional (M, D) table of the mean normalised token row of every image of a resident store
   * (ep_token_xhat_mean over the whole store, same ln eps; indexed like x through image_index; needs token_stats).  It
   * depends on the frozen tokens only: with it a step reads the batch's tokens twice instead of three times.  NULL: computed. */
  const float* xhat_mean;
} ep_clip_step;

int64_t ep_clip_head_param_offsets(const ep_clip_dims* dims, int64_t offsets[9]);
size_t ep_clip_head_workspace_bytes(const ep_clip_dims* dims);
int ep_clip_head_train_step(const ep_clip_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_clip_head_eval_forward(const ep_clip_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                              const int32_t* image_index, const float* token_stats, const float* params,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                              void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DOLG spatial attention pooling (reference poolings/dolg/dolg.py:11-62 SpatialAttention2d as the registry builds it,
 * probe_heads.py:82: SpatialAttention2d(in_c=dim, s3_dim=dim, with_aspp=False)):
 *     Yh = BatchNorm2d(conv1(x)) over the square token grid ; F = Yh / max(|Yh|_2, 1e-12) per token ;
 *     att = softplus(conv2(relu(Yh))) ; out[b] = mean_n att[b,n] F[b,n]
 * Matrix-core bound (one (B N) x D x D contraction per direction).  Six tensors: conv1.weight (D,D,1,1) .bias | bn.weight
 * .bias | conv2.weight (1,D,1,1) .bias (1); the BatchNorm's running_mean / running_var / num_batches_tracked are buffers.
 * The tokens must be a dense fp32 (B, N, D) tensor with N a perfect square (as the reference's view(b, c, h, w) needs).
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_dolg_dims {
  int32_t B, N, D, C;
} ep_dolg_dims;

typedef struct ep_dolg_params {
  float *conv1_w, *conv1_b, *bn_w, *bn_b, *conv2_w, *conv2_b;
} ep_dolg_params;

size_t ep_dolg_pool_workspace_bytes(const ep_dolg_dims* dims);
int ep_dolg_pool_forward(const ep_dolg_dims* dims, const void* x, int x_dtype, int64_t x_bstride, int training, float bn_eps,
                         float bn_momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         const ep_dolg_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_dolg_pool_backward(const ep_dolg_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const ep_dolg_params* params, const float* dy, const ep_dolg_params* grads, int accumulate,
                          void* ws, size_t ws_bytes, ep_stream_t stream);
/* attention scores (B, N) of the last forward on this workspace (dolg.py return_attn) */
int ep_dolg_attention(const ep_dolg_dims* dims, const void* ws, float* att, ep_stream_t stream);

typedef struct ep_dolg_step {
  ep_dolg_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;                        /* must be NULL (gather the batch first) */
  const float* image_stats;                          /* unused; layout parity with ep_aim_step */
  float* tok_running_mean; float* tok_running_var; int64_t* tok_num_batches_tracked;   /* 0.bn.* buffers */
  float tok_bn_eps, tok_bn_momentum;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_dolg_step;

/* flat layout: conv1.weight | conv1.bias | bn.weight | bn.bias | conv2.weight | conv2.bias | fc.weight | fc.bias */
int64_t ep_dolg_head_param_offsets(const ep_dolg_dims* dims, int64_t offsets[8]);
size_t ep_dolg_head_workspace_bytes(const ep_dolg_dims* dims);
int ep_dolg_head_train_step(const ep_dolg_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_dolg_head_eval_forward(const ep_dolg_dims* dims, const void* x, int x_dtype, int64_t x_bstride, float tok_bn_eps,
                              const float* tok_running_mean, const float* tok_running_var, const float* params,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                              void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CBAM pooling (reference poolings/cbam.py:104-139 CbamPooling with ChannelAttn :19-36 and SpatialAttn :57-68 as the registry
 * builds it, probe_heads.py:77: CbamPooling(channels=dim, spatial_kernel_size=7) -> reduction 1/16 (rd = int(D / 16 + 0.5)),
 * no MLP bias, sigmoid gates, 7x7 convolution + one-channel BatchNorm2d on the square token grid):
 *     out[b,c] = mean_n relu(x gc gs + x) = R0[b,c] + gc[b,c] mean_n gs[b,n] relu(x[b,n,c])      (both gates lie in (0, 1))
 * HBM-bound: streaming passes over the frozen tokens (csrc/ep_cbam.hip).  Five tensors: channel.fc1.weight (rd,D,1,1) |
 * channel.fc2.weight (D,rd,1,1) | spatial.conv.conv.weight (1,2,ks,ks) | spatial.conv.bn.weight .bias (1); the BatchNorm's
 * running statistics are buffers.  ep_cbam_channel_table: per-image {mean_n x, max_n x, mean_n relu(x)} per channel ((B, 3, D));
 * it depends on the frozen tokens only, so a resident store computes it once and passes it as `channel_table` (rows addressed
 * through image_index like the tokens); NULL: computed for the batch (one more streaming read).
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_cbam_dims {
  int32_t B, N, D, C, rd, ks;
} ep_cbam_dims;

typedef struct ep_cbam_params {
  float *fc1_w, *fc2_w, *conv_w, *bn_w, *bn_b;
} ep_cbam_params;

int ep_cbam_channel_table(const void* x, int x_dtype, int64_t x_bstride, const int32_t* image_index, int B, int N, int D,
                          float* table, ep_stream_t stream);
size_t ep_cbam_pool_workspace_bytes(const ep_cbam_dims* dims);
int ep_cbam_pool_forward(const ep_cbam_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                         const int32_t* image_index, const float* channel_table, int training, float bn_eps,
                         float bn_momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         const ep_cbam_params* params, float* y, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_cbam_pool_backward(const ep_cbam_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                          const int32_t* image_index, const ep_cbam_params* params, const float* dy,
                          const ep_cbam_params* grads, int accumulate, void* ws, size_t ws_bytes, ep_stream_t stream);

typedef struct ep_cbam_step {
  ep_cbam_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int32_t* image_index;
  const float* image_stats;                          /* optional cached (M, 3, D) table of ep_cbam_channel_table */
  float* tok_running_mean; float* tok_running_var; int64_t* tok_num_batches_tracked;   /* 0.spatial.conv.bn.* buffers */
  float tok_bn_eps, tok_bn_momentum;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
  ep_stream_t aux_stream;
} ep_cbam_step;

/* flat layout: channel.fc1.weight | channel.fc2.weight | spatial.conv.conv.weight | bn.weight | bn.bias | fc.weight | fc.bias */
int64_t ep_cbam_head_param_offsets(const ep_cbam_dims* dims, int64_t offsets[7]);
size_t ep_cbam_head_workspace_bytes(const ep_cbam_dims* dims);
int ep_cbam_head_train_step(const ep_cbam_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_cbam_head_eval_forward(const ep_cbam_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                              const int32_t* image_index, const float* channel_table, float tok_bn_eps,
                              const float* tok_running_mean, const float* tok_running_var, const float* params,
                              const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                              void* ws, size_t ws_bytes, ep_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DINOv2-block pooling (reference poolings/other_pool.py:299-318 DinoViTBlockPooling: one poolings/dinov2_layers/block.py:43
 * Block -- LayerNorm(eps 1e-5) -> 8-head self-attention (attention.py:37, qkv without bias) -> residual -> LayerNorm ->
 * GELU MLP x4 -> residual -- and the mean over the tokens; registry entry probe_heads.py:80):
 *     x1 = x + proj(MHSA(norm1(x))) ; x2 = x1 + fc2(gelu(fc1(norm2(x1)))) ; out[b] = mean_n x2[b,n]
 * Matrix-core bound (24 N D^2 + 4 N^2 D FLOP per image forward).  Eleven tensors: norm1.weight .bias | attn.qkv.weight
 * (3D, D) | attn.proj.weight .bias | norm2.weight .bias | mlp.fc1.weight (hidden, D) .bias | mlp.fc2.weight (D, hidden)
 * .bias.  The tokens must be a dense fp32 (B, N, D) tensor; D % H == 0 and D / H, hidden multiples of 4.
 * ------------------------------------------------------------------------------------------ */
typedef struct ep_dinovit_dims {
  int32_t B, N, D, H, hidden, C;
  float ln_eps;                                      /* 1e-5 (block.py:50 norm_layer = nn.LayerNorm) */
} ep_dinovit_dims;

typedef struct ep_dinovit_params {
  float *n1_w, *n1_b, *qkv_w, *proj_w, *proj_b, *n2_w, *n2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} ep_dinovit_params;

size_t ep_dinovit_pool_workspace_bytes(const ep_dinovit_dims* dims);
int ep_dinovit_pool_forward(const ep_dinovit_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                            const ep_dinovit_params* params, float* out, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_dinovit_pool_backward(const ep_dinovit_dims* dims, const void* x, int x_dtype, int64_t x_bstride,
                             const ep_dinovit_params* params, const float* dout, const ep_dinovit_params* grads, int accumulate,
                             void* ws, size_t ws_bytes, ep_stream_t stream);
/* attention weights (B, H, N, N) of the last forward on this workspace (block.py:91 return_attention) */
int ep_dinovit_attention(const ep_dinovit_dims* dims, const void* ws, float* A, ep_stream_t stream);

typedef struct ep_dinovit_step {
  ep_dinovit_dims dims;
  const void* x; int32_t x_dtype; int64_t x_bstride;
  const int64_t* targets;
  float* params; float* grads; float* opt_state0; float* opt_state1;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* stats;
  int32_t* found_inf; float* grad_norm;
  float bn_eps, bn_momentum;
  float grad_scale, inv_scale;
  int32_t accumulate;
  int32_t optimizer;
  float lr, weight_decay, momentum, trust_coefficient, beta1, beta2, adam_eps;
  int64_t opt_step;
  int32_t phases;
} ep_dinovit_step;

/* flat layout: the eleven tensors above in that order | fc.weight | fc.bias */
int64_t ep_dinovit_head_param_offsets(const ep_dinovit_dims* dims, int64_t offsets[13]);
size_t ep_dinovit_head_workspace_bytes(const ep_dinovit_dims* dims);
int ep_dinovit_head_train_step(const ep_dinovit_step* step, void* ws, size_t ws_bytes, ep_stream_t stream);
int ep_dinovit_head_eval_forward(const ep_dinovit_dims* dims, const void* x, int x_dtype, int64_t x_bstride, const float* params,
                                 const float* running_mean, const float* running_var, float bn_eps, float* logits, int ldl,
                                 void* ws, size_t ws_bytes, ep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EP_HIP_H */
