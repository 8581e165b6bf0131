// Group 3 of the chain kernels' width triples (chain_widths.h): the backward launches.
#define WMZ_CHAIN_GROUP 3
#include "layer_chain_bwd.hip"
