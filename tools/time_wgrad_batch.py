"""The fused path's five weight gradients of a layer as ONE launch pair (wmz_linear_wgrad_batch) at config-4 sizes."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(0)
M = 65536
shapes = [(256, 256), (256, 256), (256, 128), (128, 256), (512, 256)]       # ff2, ff1, to_out, to_q, to_k|to_v
probs = []
for N, K in shapes:
    dc = torch.randn(M, N, device='cuda').bfloat16()
    a = torch.randn(M, K, device='cuda').bfloat16()
    probs.append((dc, a, torch.zeros(N, K, device='cuda'), torch.zeros(N, device='cuda'), False))
fn = lambda: ops.linear_wgrad_batch(probs)
for _ in range(5): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): fn()
g.replay(); torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1000 / 20)
byt = sum(M * (n + k) * 2 for n, k in shapes)
print(f'wgrad batch of {len(shapes)} (GEMM + reduction): {best:.1f} us per launch pair, {byt / 1e6:.0f} MB of operands = {byt / best / 1e6:.2f} TB/s')
