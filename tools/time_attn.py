"""Per-launch time of the 16-wide-plane attention forward at config-4 shapes (graph of 50 launches).
A/B between kernel builds: tools/build_variant.py <tag> attn_fwd_row16.hip -DWMZ_ATTN_MODE=<0..15>, then
WMZ_LIB_PATH=tools/variants/libwmz_<tag>.so python tools/time_attn.py (the product library carries one instantiation)."""
import os, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
qkv = torch.randn(8, 32, 16, 16, 384, device='cuda').bfloat16()
q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
ext = tuple(int(e) for e in os.environ.get('WMZ_EXT', '3,3,3').split(','))
for _ in range(400): ops.local3d_attention_fwd(q, k, v, ext, 1)          # clocks / caches settled before the first timing
torch.cuda.synchronize()
for var in [os.environ.get('WMZ_LIB_PATH', 'product library')]:
    for _ in range(5): ops.local3d_attention_fwd(q, k, v, ext, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.local3d_attention_fwd(q, k, v, ext, 1)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 50)
    print(f'attn fwd variant {var} ext {ext}: {best:.2f} us per launch', flush=True)
