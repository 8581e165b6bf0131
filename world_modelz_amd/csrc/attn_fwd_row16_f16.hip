// The local 3D attention row kernel with IEEE-half MFMA operands: the precise fused inference mode (include/wmz.h, WMZ_F16).
// Same source as attn_fwd_row16.hip, compiled with the translation unit's 16-bit operand format switched (wmz_common.h: the
// conversions and the three MFMA shapes are the only places a kernel touches the VALUE of a 16-bit element); the dispatcher of
// this unit is wmz_attn_fwd_row16_dispatch_f16 (attn_fwd.hip routes dtype WMZ_F16 to it).
#define WMZ_OP16_F16 1
#include "attn_fwd_row16.hip"
