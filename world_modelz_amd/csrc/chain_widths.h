// Width triples the per-token chain kernels (layer_chain.hip, layer_chain_bwd.hip) are instantiated for:
//   (D = --dim, I = --heads x --dim_head, M = --mlp_dim, MC = hidden units per feed-forward chunk; MC D % 8192 == 0: a chunk's two
//   GEMMs are whole weight slabs).
// The templates are generic in the widths (D, I, MC multiples of 32; D <= 512 with I <= 128, else D <= 384: the fp32 stream, D / 4
// registers per lane, and the packed operands must fit 256 VGPRs); a triple costs eleven kernels (three inference launches in
// bfloat16 and in half, three training launches, two backward launches), so the table is what the reference publishes or runs
// itself plus the neighbours of the argparse defaults (main.py:176-182: --dim 256 --mlp_dim 256 --dim_head 128 --heads 1 is the
// default-width kernel, layer_fused.hip).  One translation unit per group and operand format (build time: the units compile side
// by side).
#pragma once

#define WMZ_CHAIN_WIDTHS_0(X) /* the reference's published runs, results/README.md:7-22 */ \
  X(96, 128, 256, 256) X(384, 128, 512, 64)
#define WMZ_CHAIN_WIDTHS_1(X) /* the reference's own test(), local_3d_attention.py:166-174: dim 128, 3 heads of 64, mlp 256 */ \
  X(128, 192, 256, 64) X(128, 128, 256, 64) X(128, 128, 512, 64)
#define WMZ_CHAIN_WIDTHS_2(X) \
  X(192, 128, 512, 128) X(256, 128, 512, 32) X(256, 128, 1024, 32)
#define WMZ_CHAIN_WIDTHS_3(X) \
  X(256, 256, 256, 32) X(256, 256, 512, 32) X(256, 256, 1024, 32) X(512, 128, 1024, 32)

#define WMZ_CHAIN_ALL_WIDTHS(X) WMZ_CHAIN_WIDTHS_0(X) WMZ_CHAIN_WIDTHS_1(X) WMZ_CHAIN_WIDTHS_2(X) WMZ_CHAIN_WIDTHS_3(X)
#define WMZ_CHAIN_ALL_GROUPS(G) G(0) G(1) G(2) G(3)

#define WMZ_CHAIN_CAT_(a, b) a##b
#define WMZ_CHAIN_CAT(a, b) WMZ_CHAIN_CAT_(a, b)
#define WMZ_CHAIN_WIDTHS_OF(g) WMZ_CHAIN_CAT(WMZ_CHAIN_WIDTHS_, g)
