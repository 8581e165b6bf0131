"""compute-only (dbg 2) timing of one attention-kernel build (WMZ_LIB_PATH selects a compile-time ablation variant)."""
import os, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
r = torch.randn(3, 8, 32, 16, 16, 128, device='cuda').bfloat16()
def timeit():
    for _ in range(200): ops.local3d_attention_fwd(r[0], r[1], r[2], (3, 3, 3), 1)
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
    return best
out = []
for dbg in (2, 0):
    L.call('wmz_debug_attn_knobs', dbg, 0)
    out.append(timeit())
L.call('wmz_debug_attn_knobs', 0, 0)
print(f"{os.environ.get('TAG', 'base'):28s} compute only {out[0]:6.2f} us   with staging {out[1]:6.2f} us", flush=True)
