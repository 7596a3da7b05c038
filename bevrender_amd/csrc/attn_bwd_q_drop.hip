// Region query-side backward with attention dropout: attn_bwd_q.hip compiled with the keep-mask block in (see
// attn_fwd_drop.hip).
#define BEVR_DROP 1
#include "attn_bwd_q.hip"
