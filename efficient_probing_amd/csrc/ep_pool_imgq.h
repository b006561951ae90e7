// Token passes with PER-IMAGE query rows and per-image query gradients (gfx950 / CDNA4).
//
// SimPool-style heads (reference poolings/simpool.py) derive their query from the image itself (the mean token), so
// the query rows differ per image and the backward needs d(query) per image, not summed over the batch as in EP.
// They are also "sliced": head h scores and pools only ITS channel slice [h*Dh, (h+1)*Dh) of every token, so all H
// heads together read each token exactly once:
//
//     s[b,h,n] = u[b, slice h] . k[b,n, slice h]        k = x            (tokstat == null)
//                                                       k = xhat = (x - mean_n) rstd_n   (LayerNorm-of-tokens scores)
//     A = softmax_n s ;   P[b, slice h] = sum_n A[b,h,n] v[b,n, slice h]   v = xhat (pool_ln) or the raw token x
//   backward (dP (B,D) given):
//     delta = dP[slice] . P[slice] ;  dA = dP[slice] . v[n, slice] ;  dS = A (dA - delta) ;
//     du[b, slice h] = sum_n dS[b,h,n] k[b,n, slice h]
//
// One workgroup per image.  A head is owned by a group of G lanes (8, 16 or the whole wave), each lane holding CPL
// 16-byte chunks of the slice, so a score is one lane-local dot product plus a G-lane DPP reduction and every lane of
// the group then holds it -- the softmax state (running max, sum) and the pooled slice live in registers, replicated
// per group.  HW waves cover the heads, TW waves split the tokens (merged through LDS at the end of the image).
// Tokens are read straight from global memory (coalesced 16 B per lane, TB tokens in flight per wave): the arithmetic
// is ~25 vector instructions per KiB, far from the issue limit, so no LDS staging is needed to stay HBM-bound.
// The backward recomputes the scores from u (one more lane-local dot product) instead of storing (B,H,N) scores.
#pragma once
#include "ep_common.h"
#include "ep_internal.h"

namespace ep {

struct ImgqParams {
  const void* x; int x_bf16; int64_t x_bstride; const int* index;
  int B, N, D, H;
  const float* u;        // (B, D) per-image query rows, scale folded in
  const float* tokstat;  // optional (M|B, N, 2) {mean, rstd}: LayerNorm-of-tokens scores
  int pool_ln;           // 1: pool xhat (needs tokstat); 0: pool the raw tokens
  float* P;              // (B, D) pooled slices
  float* ML;             // (B, H, 2): running max, sum of exponentials
  const float* dP;       // backward in (B, D)
  float* du;             // backward out (B, D)
};

bool imgq_supported(int D, int H);
int imgq_forward(const ImgqParams& p, hipStream_t st);
int imgq_backward(const ImgqParams& p, hipStream_t st);

// full-width per-image query rows on the PoolParams contract of the generic kernels (CLIP): see ep_pool_imgq.hip
bool imgqf_supported(const PoolParams& p);
int imgqf_forward(const PoolParams& p, hipStream_t st);
int imgqf_backward(const PoolParams& p, float* dq, hipStream_t st);

}  // namespace ep
