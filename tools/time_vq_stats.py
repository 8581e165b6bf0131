import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(0)
N, C, E = 65536, 1024, 64
x = torch.randn(N, E, device='cuda')
cb = torch.randn(C, E, device='cuda')
idx = torch.randint(0, C, (N,), device='cuda')
counts = torch.zeros(C, device='cuda'); dw = torch.zeros(C, E, device='cuda'); sq = torch.zeros(C, device='cuda')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
print('counts+dw      us', t(lambda: ops.vq_ema_stats(x, idx, cb, counts, dw, None)))
print('counts+dw+sqerr us', t(lambda: ops.vq_ema_stats(x, idx, cb, counts, dw, sq)))
idx2 = idx.clone(); idx2[torch.rand(N, device='cuda') < 0.3] = 7      # a dominant code
print('dominant code: counts+dw us', t(lambda: ops.vq_ema_stats(x, idx2, cb, counts, dw, None)))
