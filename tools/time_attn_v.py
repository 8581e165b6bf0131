"""Attention forward timing at config-4 shapes for a list of wmz_debug_attn_knobs variants: compute only (dbg 2: no K/V staging)
and the full kernel.   python tools/time_attn_v.py 0 16 32 ..."""
import os, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
ext = tuple(int(x) for x in os.environ.get('EXT', '3,3,3').split(','))
r = torch.randn(3, 8, 32, 16, 16, 128, device='cuda').bfloat16()
def timeit():
    for _ in range(100): ops.local3d_attention_fwd(r[0], r[1], r[2], ext, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.local3d_attention_fwd(r[0], r[1], r[2], ext, 1)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 50)
    return sorted(ts)[len(ts) // 2]
for v in [int(a) for a in sys.argv[1:]] or [0]:
    out = []
    for dbg in (2, 0):
        L.call('wmz_debug_attn_knobs', dbg, v)
        out.append(timeit())
    print(f"{os.environ.get('TAG', '')} variant {v:3d}  compute only {out[0]:6.2f} us   with staging {out[1]:6.2f} us", flush=True)
L.call('wmz_debug_attn_knobs', 0, 0)
