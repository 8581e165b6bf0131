"""Per-launch time of the row attention forward at config 4 (8 x 32 planes of 16 x 16, dh 128, window 7x7x7) for the library named by
WMZ_LIB_PATH -- the product, or a compile-time ablation of attn_fwd_row16.hip (tools/build_variant.py <tag> attn_fwd_row16.hip
-DWMZ_ATTN_ABL=<bits>: 1 no LDS fragment reads, 2 no softmax arithmetic, 4 no MFMAs; results are garbage, times are not)."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(0)
qkv = torch.randn(8, 32, 16, 16, 384, device='cuda').bfloat16()
q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
for _ in range(20): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
g.replay(); torch.cuda.synchronize()
ts = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1000 / 50)
print(f'{sys.argv[1] if len(sys.argv) > 1 else "product"}: min {min(ts):.2f} median {sorted(ts)[3]:.2f} us per launch', flush=True)
