// The plain / last-frame linear (VqVideoDiffusionModel.logit_proj, main.py:33-36) with IEEE-half MFMA operands: the precise fused
// inference mode's last step (include/wmz.h: wmz_linear_fwd_f16, wmz_linear_fwd_stats_f16, wmz_linear_fwd_blocked_f16; fp32 output).
// Same source as linear_fwd.hip with the translation unit's 16-bit operand format switched (wmz_common.h).
#define WMZ_OP16_F16 1
#include "linear_fwd.hip"
