// Region key-side backward with attention dropout: attn_bwd_k.hip compiled with the keep-mask blocks in (see
// attn_fwd_drop.hip).
#define BEVR_DROP 1
#include "attn_bwd_k.hip"
