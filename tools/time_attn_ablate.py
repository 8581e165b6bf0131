"""Ablation timing of the row16 attention forward (wmz_debug_attn_knobs): full kernel, K/V staging only (dbg 1: no
per-tile compute), compute only (dbg 2: no staging; garbage results)."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
r = torch.randn(3, 8, 32, 16, 16, 128, device='cuda').bfloat16()
def timeit(tag):
    for _ in range(100): ops.local3d_attention_fwd(r[0], r[1], r[2], (3, 3, 3), 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.local3d_attention_fwd(r[0], r[1], r[2], (3, 3, 3), 1)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 50)
    print(f'{tag}: {best:.2f} us', flush=True)
for rep in range(2):
    for dbg, tag in ((0, 'full kernel '), (1, 'staging only'), (2, 'compute only')):
        L.call('wmz_debug_attn_knobs', dbg, 0)
        timeit(tag)
L.call('wmz_debug_attn_knobs', 0, 0)
