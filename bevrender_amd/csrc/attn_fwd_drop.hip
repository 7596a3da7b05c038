// Region forward with attention dropout: the same source as attn_fwd.hip, compiled with the keep-mask block in
// (bevr_common.h: bevr_drop_keep) -- a separate translation unit so that the kernel without dropout is unchanged.
#define BEVR_DROP 1
#include "attn_fwd.hip"
