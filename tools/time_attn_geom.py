"""Attention core forward / backward per launch at a given geometry: python tools/time_attn_geom.py B S H W dh eS eH eW"""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
B, S, H, W, dh, eS, eH, eW = (int(a) for a in sys.argv[1:9])
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, S, H, W, dh, device='cuda').bfloat16() for _ in range(4))
ext = (eS, eH, eW)
out, lse, _ = ops.local3d_attention_fwd(q, k, v, ext, 1, need_lse=True)
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / reps)
    return best
f = timeit(lambda: ops.local3d_attention_fwd(q, k, v, ext, 1, need_lse=True))
b = timeit(lambda: ops.local3d_attention_bwd(q, k, v, out, lse, do, ext, 1))
print(f'attention core B={B} {S}x{H}x{W} dh={dh} ext={ext}: forward {f:.1f} us, backward (both passes) {b:.1f} us')
