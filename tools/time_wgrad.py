"""Weight-gradient GEMM timing at config-4 sizes (M = 65536 tokens): two-stage (workspace) vs float-atomic split-K."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
M = 65536
def timeit(fn, tag):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 20)
    print(f'{tag}: {best:.1f} us', flush=True)
for N, K in ((256, 256), (128, 256), (256, 128), (1024, 256)):
    Mx = M if N != 1024 else 2048
    dc = torch.randn(Mx, N, device='cuda').bfloat16()
    a = torch.randn(Mx, K, device='cuda').bfloat16()
    dw = torch.zeros(N, K, device='cuda')
    db = torch.zeros(N, device='cuda')
    mean, rstd = ops.layernorm_stats(a)
    g, b = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    timeit(lambda: ops.linear_wgrad(dc, a, dw, db), f'wgrad {N}x{K} M={Mx} two-stage plain ')
    timeit(lambda: ops.linear_wgrad(dc, a, dw, db, ln=(g, b), ln_stats=(mean, rstd)), f'wgrad {N}x{K} M={Mx} two-stage LN    ')
    timeit(lambda: ops.linear_wgrad(dc, a, dw, db, gelu_in=True), f'wgrad {N}x{K} M={Mx} two-stage GELU  ')
    timeit(lambda: L.call('wmz_linear_wgrad', L.ptr(dc), N, L.ptr(a), K, L.ptr(dw), L.ptr(db), Mx, N, K, None, None, None, None, 0, 1, L.stream()),
           f'wgrad {N}x{K} M={Mx} atomics plain   ')
