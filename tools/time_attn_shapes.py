import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(0)
def run(B, S):
    qkv = torch.randn(B, S, 16, 16, 384, device='cuda').bfloat16()
    q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
    for _ in range(50): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 50)
    print(f'B {B} S {S}: {best:.2f} us per launch ({B * S} planes)', flush=True)
for B, S in ((8, 32), (8, 16), (16, 16), (8, 8), (32, 8), (4, 32), (2, 32)):
    run(B, S)
