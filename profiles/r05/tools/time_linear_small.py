"""The small-M nn.Linear GEMMs of config 5 (3 072 rows per GPU) and of the last-frame logits: the LDS-DMA ring K loop against the
register-staged one (wmz_debug_linear_knobs), 20 launches in a hipGraph each; results compared."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    for _ in range(3):
        g.replay()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / 20)
    return best


torch.manual_seed(0)
for (M, N, K) in ((3072, 512, 512), (3072, 1536, 512), (3072, 1024, 512), (3072, 512, 1024), (2048, 1024, 256), (3072, 8192, 512)):
    x = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda').bfloat16()
    out = {}
    for dma in (1, 0):
        L.call('wmz_debug_linear_knobs', dma)
        y = ops.linear_fwd(x, w, bias, residual=res)
        out[dma] = (y, timed(lambda: ops.linear_fwd(x, w, bias, residual=res)))
    L.call('wmz_debug_linear_knobs', 1)
    same = torch.equal(out[0][0], out[1][0])
    ref = (x.float() @ w.float().t() + bias + res.float())
    err = float((out[1][0].float() - ref).norm() / ref.norm())
    print(f'M={M} N={N} K={K}: DMA ring {out[1][1]:6.1f} us, register-staged {out[0][1]:6.1f} us, identical {same}, rel err vs fp32 {err:.1e}', flush=True)
