// Group 1 of the chain kernels' width triples (chain_widths.h): the backward launches.
#define WMZ_CHAIN_GROUP 1
#include "layer_chain_bwd.hip"
