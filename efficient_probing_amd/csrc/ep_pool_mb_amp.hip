// The single-product (AMP-bf16 arithmetic, NT = 1) instances of the bf16-token matrix-core passes: the kernels are the templates of
// ep_pool_mb.hip (mb2 family); compiled here so that the two sets of instances build in parallel (ep_pool_mb.hip alone is the
// longest compile of the library).  Entry: mb2_launch_amp(), called by mb2_launch_one() when PoolParams.nterms == 1.
#define EP_MB_AMP_TU 1
#include "ep_pool_mb.hip"
