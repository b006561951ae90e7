// Internal (non-ABI) interfaces between the translation units of libep_hip.so.
#pragma once
#include "ep_common.h"

namespace ep {

struct PoolParams {
  const float* x;        // tokens (fp32, or bf16 when x_bf16: then this pointer is only ever used as a byte address)
  int x_bf16;            // 1: tokens are stored as bf16 (widened to fp32 on load; all arithmetic stays fp32)
  int64_t x_bstride;     // elements between images
  int B, N, D, Q;
  int nterms;            // bf16-token matrix-core passes (ep_pool_mb.hip: mb2): 0 / 3 = the fp32 operand as three bf16 terms (fp32 results);
                         // 1 = AMP-bf16 arithmetic: the operand rounded to bf16, ONE product.  Other kernel families ignore it.
  int Qs;                // queries per image in MEMORY (batch stride of P / S / ML / dP in query rows); 0 = Q.  Larger than Q
                         // when a launch covers a chunk of the queries (ep_pool.hip: chunked passes for Q beyond a kernel family's limit)
  const float* cls;      // fwd: (Q,D) or (B,Q,D)
  int64_t cls_bstride;   // 0 when shared
  float scale;
  float* P;              // fwd out (B,Q,D)
  float* S;              // fwd out / bwd in (B,Q,N)
  float* ML;             // (B,Q,4)
  const float* dP;       // bwd in (B,Q,D)
  float* Gpart;          // bwd out (n_workgroups, Q, D)
  const int* index;      // optional (B,) image indices into x: image b is x[index[b]] (resident token store)
  const float* tokstat;  // optional (M or B, N, 2) per-token {mean, rstd} of a LayerNorm over D (indexed like x): the
                         // pass then pools the NORMALISED tokens xhat = (x - mean) * rstd without materialising them:
                         // scores q.xhat, pooled sum_n A xhat; backward accumulates sum dS xhat
  // per-(image, query, token) extras, honoured by the generic kernels and the full-width per-image-query kernel
  // (ep_pool_imgq.hip) only (setting any of them selects those):
  const float* sbias;    // fwd: added to the score before the softmax (the stored S includes it)   (B,Q,N)
  const float* dabias;   // bwd: added to dA = dP . v_n                                               (B,Q,N)
  float* dSout;          // bwd: the score gradients dS                                               (B,Q,N)
  // bwd, vector-ALU streaming kernel only: the softmax-correction term delta[b,q] = sum_c dyv[b, q Dv/Q + c] yv[b, q Dv/Q + c]
  // (= dP[b,q] . P[b,q]) computed by the pass itself from one extra ring item per image instead of being read from
  // ML[b,q,2] (saves the ep_delta_kernel launch in front of the pass); null: read ML
  const float* dyv; const float* yv; int Dv;
  // ---- in-pass contractions of the fused EP step (ep_inpass.h; 4-wave vector-ALU streaming kernels, Q = 8, D = 256 KP,
  // d_out = 1, B and the pooling grid multiples of 32).  Image b owns "task b" of its 32-image row block: one of the
  // 8 queries x 4 column (K) quarters.  Counters: one int per row block, ZERO at launch, 4 wave arrivals per task.
  // second pass: dP[b, q, :] = dy[b, q-slice] Wv[q-slice, :] is produced by the pooling workgroups themselves before they
  // stream (no ep_gemm launch in front of the pass): ip_dy (B, D), ip_Wv (D, D), the result goes to the buffer `dP` points at
  const float* ip_dy; const float* ip_Wv; int* ip_dcnt;
  // ... with the BatchNorm backward folded into those tasks (ip_fold_dz != null): dy = rstd (dz - m1 - z m2) is formed while a
  // task stages its A tile, m1 / m2 from the per-tile column statistics the dz contraction left (GemmParams.cs_out); the
  // tasks of column quarter 0 also WRITE dy (ip_dy) for the delta items and the dWv side tasks of the same launch
  const float* ip_fold_dz; const float* ip_fold_z; const float* ip_fold_rstd; const float* ip_fold_cs;
  // first pass: y = P_q Wv_q^T as four K-quarter partials ip_ypart[ks][b][:] by the workgroups that have finished their
  // images (the matrix pipe is idle under the pass); BatchNorm sums the partials in fixed order
  const float* ip_WvF; float* ip_ypart; int* ip_ycnt;
  float* ip_y; int ip_yr0;      // rows < ip_yr0 (row blocks whose images end before the last round) get their y directly, all
                                // four K quarters summed in one task by a workgroup that is done early; rows >= ip_yr0 as partials
  int* ip_zero; int ip_nzero;   // counters this launch clears for the OTHER pass (first instructions of workgroup 0)
  // ticketed second pass (ep_pool_bwd2.hip): images are handed out by a device-wide counter (zero at launch); workgroups
  // with index < tick_base stream image `index` first, a ticket t names image tick_base + t
  int* tick; int tick_base;
  int* ip_err;           // bumped when a bounded flag wait gives up (never in a correct run; tests read it)
  int nslot;             // ring depth
  int slot_bytes;        // TT*D*4
  int kdma;              // 16-byte DMA instructions per wave per ring item
  unsigned long long* dbg;  // diagnostic only (EP_MF_STAMP): per-wave phase cycle counters
  int ablate;            // diagnostic only (EP_POOL_ABLATE): 1 ring only, 2 +LDS reads, 3 no pooling FMAs
};

struct GemmParams {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;            // leading dimensions (elements)
  int64_t sAz, sBz, sCz;            // batch strides (elements)
  int64_t sBiasz;                   // batch stride of the bias vector (0: shared)
  int extA, extB;                   // readable extent of the contiguous dim of a T-layout operand
  float alpha;
  int accumulate;
  int side;                         // 1: runs on the aux stream beside a token pass (kernel choice hint)
  float* skws; size_t skws_floats;  // optional scratch for a split of K (few output tiles, very long K): >= 2 M N floats
  int ablate;                       // diagnostic only
  int npers;                        // LDS-DMA kernel: N-tiles one workgroup walks (0 / 1: one; set by gemm())
  // 32 x 96-tile K/T kernel only (gemm_colstats_ok): column statistics of every 32-row tile of C in its epilogue, for a
  // BatchNorm backward folded into C's consumer (ep_inpass.h): cs_out[(tile row * 2 + 0) * N + col] = sum_rows C,
  // [.. + 1 ..] = sum_rows C * cs_z (cs_z: M x N, leading dimension ldc)
  const float* cs_z; float* cs_out;
  // ep_planes.hip: the B operand as pre-split bf16 planes (weights; see planes_split) -- [term][row][ldbp], K contiguous
  const uint16_t* Bpl; int64_t pl_term, ldbp, sBpz;   // plane base, term stride, row stride, batch offset (elements)
  int nterms;                       // bf16 matrix-core kernels (planes, b3): 0 / 3 = three-term split at fp32 accuracy; 1 = ONE product of the
                                    // operands rounded to bf16, fp32 accumulation (the AMP-bf16 arithmetic mode: gemm_arith())
  int zn;                           // planes kernel: batch count of a 1-D (m_fast) launch, set by gemm_planes
  // BatchNorm1d(affine=False) in EVAL mode folded into the epilogue (round 6, ep_head_eval_forward: the value projection writes
  // z = (y - running_mean[col]) / sqrt(running_var[col] + eps) directly -- the arithmetic of ep_bn_eval_kernel, one launch less);
  // indexed by the output column like `bias` (batched launches: z * sBiasz + col)
  const float* bn_rm; const float* bn_rv; float bn_eps;
  // bf16 x3 tile (ep_wgrad3.h), K split INSIDE one launch (round 6, small weight gradients on side queues): ksplit > 1 makes batch
  // entry bz = zb * ksplit + ks contract slice ks of the K range -- operands offset by ks * ksA / ks * ksB elements, K = the slice
  // length -- into the partial matrix C + zb * sCz + ks * ksC; the caller sums the slices in order (reduce_partials)
  int ksplit; int64_t ksA, ksB, ksC;
  int m_fast;                       // planes kernel: M-tiles fastest in the launch order (few M-tiles against a very long N: the
                                    // workgroups that share a weight tile then run together and it is fetched from HBM once)
};

// a row-major weight matrix W (R x K, leading dimension ldw) and where its planes go (ep_planes.hip); either may be null
struct PlaneSpec {
  const float* W; int R, K; int64_t ldw;
  uint16_t* pn;                     // planes of W   : 3 * R * round_up(K, 32) elements
  uint16_t* pt;                     // planes of W^T : 3 * K * round_up(R, 32) elements
};
// Arithmetic mode of the calling thread's contractions (ep_gemm.hip): 0 = fp32 results (default), 1 = AMP-bf16 -- every
// contraction on the bf16 matrix-core kernels as ONE product of bf16-rounded operands with fp32 accumulation (what the
// reference's --amp bfloat16 does inside autocast, engine_finetune.py:52-55, minus the bf16 rounding of the outputs).
// Set for the duration of a train step by ep_head_train_step (ep_head_step.arith).
int gemm_arith();
void gemm_set_arith(int a);
// sets the mode for the calls a step function enqueues, restores the previous one on every return path
struct ArithScope {
  int old;
  explicit ArithScope(int a) : old(gemm_arith()) { gemm_set_arith(a); }
  ~ArithScope() { gemm_set_arith(old); }
  ArithScope(const ArithScope&) = delete;
  ArithScope& operator=(const ArithScope&) = delete;
};
size_t planes_elems(int rows, int K);               // elements (uint16) of the planes of a rows x K matrix
int planes_split(const PlaneSpec* specs, int n, hipStream_t st);
bool gemm_planes_ok(const GemmParams& p);
// C[z] (+)= alpha * A[z] W[z]^T (+ bias): A fp32 with K contiguous (lda), W given as planes (Bpl ...); fp32 accuracy
int gemm_planes(const GemmParams& p, int batch, hipStream_t st);
struct SideTasks;
// ... with side tasks (ep_sidetask.h: weight-gradient contractions, column sums, the statistics fold) as extra workgroups of the launch
int gemm_planes_side(const GemmParams& p, int batch, const SideTasks& sd, hipStream_t st);
bool planes_big_wanted(const GemmParams& p, int batch);      // ep_planes_big.hip: 128 x 128 tiles (large contractions)
void planes_big_launch(const GemmParams& p, int batch, hipStream_t st);
// dst (C x R, ldd) = src (R x C, lds_)^T, fp32 (ep_planes.hip)
int transpose_f32(const float* src, int R, int C, int64_t lds_, float* dst, int64_t ldd, hipStream_t st);

// Work appended to the launch of the second token pass (ep_side.h: run_side_task)
struct SideTasks {
  GemmParams g[6];                  // T/T-layout, 16-byte aligned operands
  int gx[6], gy[6], gz[6], bm[6];   // tile grid and tile rows (32 / 64) of each; gx*gy*gz == 0 when unused
  int n_gemm;
  const float* cs_src; float* cs_out; int cs_B, cs_ncol, cs_ld, cs_accumulate, n_colsum;
  const float* rowstat; float* stats; int rs_B, n_stats;
  // further column sums (bias gradients of the heads' dense layers: out[c] (+)= sum_b src[b][c]) -- side_add_colsum
  const float* xcs_src[4]; float* xcs_out[4]; int xcs_B[4], xcs_ncol[4], xcs_ld[4], xcs_acc[4], xcs_blocks[4]; int n_xcs;
  int total;                        // number of extra workgroups
  int first_block;                  // set by the launcher: side workgroups occupy blocks [first_block, first_block + total)
  int xcd_order;                    // 1: tiles of a contraction are handed out XCD by XCD (run_side_task)
  int b3;                           // 1: the contractions run on the bf16 x3 tile (ep_wgrad3.h) instead of the exact-f32 one
};

// element offset of image b of the batch inside the token buffer
#define EP_IMG_OFF(p, b) ((int64_t)((p).index ? (p).index[(b)] : (b)) * (p).x_bstride)

size_t pool_workspace_bytes(int B, int N, int D, int Q);
const char* pool_kernel_family(int B, int N, int D, int Q, int bwd, int x_bf16 = 0);
int pool_forward(const PoolParams& p, hipStream_t st);
// `side` (optional): extra work to run inside the launch; honoured only when pool_backward_takes_side(p)
// The last stage of a two-stage partial reduction, handed to the optimizer instead of being launched: element j of
// out[0..n) = (accumulate ? out[j] : 0) + scale * sum_{r < 16} stage[r n + j], summed in ep_reduce_partials_kernel's order.
struct DeferredReduce {
  const float* stage;               // null: nothing deferred (the reduction ran in full)
  float* out; int n; float scale; int accumulate;
};
// `defer` (optional): leave the final 16 -> 1 stage of the dcls reduction to optim_step (same call, same stream)
int pool_backward(const PoolParams& p, float* dcls, int accumulate, hipStream_t st, const SideTasks* side = nullptr,
                  DeferredReduce* defer = nullptr);
bool pool_backward_takes_side(const PoolParams& p);
bool pool_backward_takes_delta(const PoolParams& p, int Dv);   // ... and compute the delta rows itself (dyv / yv / Dv)
// In-pass contractions (ep_inpass.h) possible for this shape on BOTH passes?  bit 0: y inside the first pass, bit 1: dP
// inside the second (EP_INPASS=<mask> switches them; default 2: dP inside the second pass only).  Dv = width of the projection (must equal D).
int pool_inpass_mask(const PoolParams& p, int Dv);
struct StreamGridInfo { int grid, rounds, helpers; };          // pooling workgroups, ceil(B / grid), workgroups without an image of the last round
StreamGridInfo pool_stream_grid(const PoolParams& p);
// bit 2 of that mask: the ticketed second pass (ep_pool_bwd2.hip: dP and the weight-gradient side tasks under the stream)
int bwd2_launch(const PoolParams& p, int grid, int first, hipStream_t st, const SideTasks* side);
constexpr int IP_YPARTS = 4;                                   // K quarters of the in-pass value projection
// per-image query gradients: dq (B,Q,D) = p.scale * sum_n dS[b,q,n] k[b,n,:], NOT summed over the batch (per-image query rows)
int pool_backward_per_image(const PoolParams& p, float* dq, hipStream_t st);
bool gemm_side_ok(const GemmParams& p, bool a_k, bool b_k);
// one or two SMALL T / T weight gradients (C (+)= A^T B, batched) in one launch of the paired-group tile (ep_wgrad3.h: gemm_tile_b3p)
bool wgrad_pair_ok(const GemmParams& p);
int wgrad_pair(const GemmParams* g, const int* batch, int n, hipStream_t st);
const char* gemm_kernel_name(bool a_k, bool b_k, const GemmParams& p, int batch);   // what gemm() would launch (family name)
bool gemm_b3_on();                                             // EP_GEMM_B3 (default 1): T / T contractions on the bf16 x3 tile
int debug_force_generic(int on);
int token_stats(const void* x, int x_bf16, int64_t bstride, int B, int N, int D, float eps, float* stats, hipStream_t st);
int attention_from_scores(const float* S, const float* ML, int rows, int N, float* A, hipStream_t st);

// a_k / b_k: operand contiguous along K (true) or along its free dimension (false)
int gemm(bool a_k, bool b_k, const GemmParams& p, int batch, hipStream_t st);
bool gemm_colstats_ok(bool a_k, bool b_k, const GemmParams& p, int batch);   // will gemm() run the kernel that honours cs_out?

size_t bn_workspace_bytes(int B, int Dp);
// nparts > 1 (one-launch kernel only, B <= 1024): y is given as nparts partial matrices `pstride` floats apart, summed in
// fixed order ((p0 + p1) + (p2 + p3)); the sum is also written to y_out (may be null)
int bn_forward_train(const float* y, int B, int Dp, float eps, float momentum, float* z, float* rstd,
                     float* rmean, float* rvar, int64_t* nbt, float* partial, hipStream_t st, int nparts = 1,
                     int64_t pstride = 0, float* y_out = nullptr, int r0 = 0);   // rows < r0: already summed, in y_out
// rows in front of which the in-pass projection may write y directly (0: none; the BatchNorm kernel has this one split)
int bn_parts_r0(int B, int r0);
bool bn_takes_parts(int B);
int bn_forward_eval(const float* y, int B, int Dp, float eps, const float* rmean, const float* rvar, float* z,
                    hipStream_t st);
int bn_backward(const float* dz, const float* z, const float* rstd, int B, int Dp, float* dy, float* partial,
                hipStream_t st);
int colsum(const float* src, int B, int ncol, int ld, int accumulate, float* out, hipStream_t st);
// delta[r] = sum_c dy[r,c] * (y[r,c] - bias[(r % Q)*Dq + c]) -> ML[r,2]   (bias may be null; rows = B*Q)
int delta_rows(const float* dy, const float* y, int rows, int Dq, float* ML, hipStream_t st,
               const float* bias = nullptr, int Q = 1);
// Device-resident GradScaler state (ABI v26: ep_head_step.scaler_state): two slots of {scale, growth tracker}; a step READS
// slot `slot` (cross entropy: grad_scale * scale; optimizer: inv_scale / scale) and its optimizer phase WRITES the other one
// (torch.cuda.amp.GradScaler.update: backoff on a non-finite gradient, growth after `interval` clean steps) -- no host sync.
struct ScalerDev { float* state; int slot; float growth, backoff; int interval; };
// scale_dev (optional): one device float the gradient scale is multiplied with
int cross_entropy(const float* logits, int ldl, const int64_t* targets, int B, int C, float grad_scale,
                  float* loss_rows, float* dlogits, float* rowstat, hipStream_t st, const float* scale_dev = nullptr);
int ce_stats(const float* rowstat, int B, float* stats, hipStream_t st);

// shared host helpers (ep_api.hip)
int get_events(hipEvent_t* out, int n);
int check_tokens(const void* x, int x_dtype, int64_t x_bstride, int B, int N, int D, int Q);
PoolParams pool_params(const void* x, int64_t x_bstride, int B, int N, int D, int Q, float scale, int x_dtype = 0);
int linear_forward(const float* z, const float* Wc, const float* bc, int B, int Dp, int C, float* logits, int ldl,
                   hipStream_t st);
int linear_backward(const float* dl, int ldl, const float* z, const float* Wc, int B, int Dp, int C, float* dz,
                    float* dWc, float* dbc, int accumulate, hipStream_t st);
GemmParams dwc_gemm(const float* dl, int ldl, const float* z, int B, int Dp, int C, float* dWc, int accumulate);
void side_add_gemm(SideTasks& sd, const GemmParams& g, int batch);
// a bias gradient nothing consumes before the optimizer: as side workgroups too (false: the list is full -- launch it)
bool side_add_colsum(SideTasks& sd, const float* src, int B, int ncol, int ld, int accumulate, float* out);
// run the side tasks as stand-alone launches on `st` (kernel families that cannot carry them)
int side_run_standalone(const SideTasks& sd, hipStream_t st);
// Side work of a head step whose token-pass kernel takes no side workgroups: on the aux stream, EARLY (round 4; EP_WGRAD_EARLY=0:
// all of it beside the second pass) -- every contraction starts as soon as its operands exist, beside the chain of small
// kernels and critical-path contractions in front of the pass that leaves most of the chip idle, instead of starting with the
// HBM-bound pass (which fills every CU) and running on after it.  Usage: begin; after producing the operands of some
// contractions: side_add_gemm(sd, ...) + fork; rest() once the column sums' sources exist; before_pass() in front of the
// pass launch; join() in front of the first consumer of a side result.  aux == st or null: everything inline on `st`.
struct AuxSide {
  hipStream_t st = nullptr, side = nullptr, side2 = nullptr;   // side2: the library's second side queue (begin(..., two = true))
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool early = false, rest_done = false, used2 = false;
  int launched = 0, nev = 0, nfork = 0;
};
// two: successive forks alternate between the aux stream and a second side queue (heads whose four or five gradient contractions
// of 45 - 80 us each queue up on ONE stream and run into the second token pass, which starves them: V-JEPA, SigLIP)
int aux_side_begin(AuxSide& a, hipStream_t st, hipStream_t aux, bool two = false);
int aux_side_fork(AuxSide& a, const SideTasks& sd);        // early: aux waits for `st` so far, then runs sd.g[launched .. n_gemm)
int aux_side_rest(AuxSide& a, const SideTasks& sd);        // early: the column sums and statistics of `sd`, now
int aux_side_before_pass(AuxSide& a, const SideTasks& sd); // whatever of `sd` has not been launched (not early: all of it)
int aux_side_join(AuxSide& a);                             // `st` waits for the aux stream
// one T/T contraction whose operands exist now: early -> on the aux stream behind everything enqueued on `st` so far; else inline
int aux_side_gemm(AuxSide& a, const GemmParams& g, int batch);
// The classifier's backward (reference probe_heads.py:76 Linear and its autograd): dz = dlogits Wc on `st`; dWc = dlogits^T z
// and dbc = column sums of dlogits feed nothing before the optimizer -- early on the aux stream (AuxSide) or inline.
int classifier_backward(AuxSide& a, const float* dlogits, int ldl, const float* z, const float* Wc, int B, int D, int C, float* dz,
                        float* dWc, float* dbc, int accumulate);
int reduce_partials(const float* parts, int nparts, int n, float scale, int accumulate, float* out, float* stage,
                    hipStream_t st, DeferredReduce* defer = nullptr);

size_t optim_workspace_bytes(int64_t total, int nseg);
int optim_step(int mode, float* p, const float* g, float* s0, float* s1, int64_t total, const ep_segment* segs,
               int nseg, float lr, float wd, float momentum, float tc, float inv_scale, float beta1, float beta2,
               float eps, int64_t step, int32_t* found_inf, float* grad_norm, void* ws, size_t ws_bytes,
               hipStream_t st, const DeferredReduce* red = nullptr,    // red: finish that reduction into g first (g is written)
               const int* abort_flag = nullptr, float* abort_stat = nullptr,   // nonzero *abort_flag: skip the update, set found_inf, bump *abort_stat
               const PlaneSpec* emit = nullptr, int n_emit = 0,    // weight matrices (whole segments) whose planes the update writes (<= 2)
               const ScalerDev* scaler = nullptr);                 // device-resident loss scale: unscale by it, update it
// how many of `emit` optim_step will really serve for this segment list (the others need planes_split)
int optim_emits(const float* p, const ep_segment* segs, int nseg, const PlaneSpec* emit, int n_emit);

}  // namespace ep
