// The per-token chain kernel of the published widths (dim 96 / mlp 256, dim 384 / mlp 512) with IEEE-half MFMA operands and a half
// stream: the precise fused inference mode on those widths (include/wmz.h: wmz_layer_chain_fwd_planes_f16).  Same source as
// layer_chain.hip with the translation unit's 16-bit operand format switched (wmz_common.h); inference only.
#define WMZ_OP16_F16 1
#include "layer_chain.hip"
