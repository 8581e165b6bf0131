"""Per-launch time of each attention backward role (MODE 0: dq, MODE 1: dk | dv) at config-4 shapes, through rocprof-free
graph timing: the two launches are timed together and separately via the general entry point's pieces."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(0)
r = torch.randn(4, 8, 32, 16, 16, 128, device='cuda').bfloat16()
q, k, v, do = r[0], r[1], r[2], r[3]
out, lse, _ = ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1, need_lse=True)
fn = lambda: ops.local3d_attention_bwd(q, k, v, out, lse, do, (3, 3, 3), 1)
for _ in range(20): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): fn()
for _ in range(3): g.replay()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1000 / 20)
print(f'attention backward (dq pass + dk|dv pass): {best:.1f} us per call')
