"""Attention forward timing at config-4 shapes for (variant, dbg) pairs of wmz_debug_attn_knobs.  python tools/time_attn_dbg.py v:d v:d ..."""
import os, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
ext = tuple(int(x) for x in os.environ.get('EXT', '3,3,3').split(','))
r = torch.randn(3, 8, 32, 16, 16, 128, device='cuda').bfloat16()
def timeit():
    for _ in range(50): ops.local3d_attention_fwd(r[0], r[1], r[2], ext, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.local3d_attention_fwd(r[0], r[1], r[2], ext, 1)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 50)
    return sorted(ts)[len(ts) // 2]
L.call('wmz_debug_attn_knobs', 0, 0)
timeit()                                   # clocks
for a in sys.argv[1:]:
    v, d = (int(x) for x in a.split(':'))
    L.call('wmz_debug_attn_knobs', d, v)
    print(f"variant {v:3d} dbg {d:3d}: {timeit():6.2f} us", flush=True)
L.call('wmz_debug_attn_knobs', 0, 0)
