// The fused per-token inference kernels (embedding + to_q | to_k | to_v; to_out + residual -> LayerNorm -> feed-forward ->
// residual -> next layer's q | k | v) with IEEE-half MFMA operands and a half residual stream between the layers: the precise
// fused inference mode (include/wmz.h: wmz_layer_fused_fwd*_f16, wmz_embed_qkv_fused_fwd*_f16, wmz_layer_fused_pack_f16,
// wmz_fused_pack_table_f16).  Same source as layer_fused.hip with the translation unit's 16-bit operand format switched
// (wmz_common.h); the training forward and the backward streams exist in the bfloat16 unit only.
#define WMZ_OP16_F16 1
#include "layer_fused.hip"
