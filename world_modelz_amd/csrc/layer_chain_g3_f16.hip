// Group 3 of the chain kernels' width triples (chain_widths.h): the inference launches with IEEE-half operands (precise mode).
#define WMZ_OP16_F16 1
#define WMZ_CHAIN_GROUP 3
#include "layer_chain.hip"
