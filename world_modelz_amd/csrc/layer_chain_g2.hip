// Group 2 of the chain kernels' width triples (chain_widths.h): the forward launches, bfloat16 unit (inference + training).
#define WMZ_CHAIN_GROUP 2
#include "layer_chain.hip"
