"""Attention forward launch time vs the memory layout of q / k / v (row strides), config-4 shapes."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(0)
B, S, H, W, I = 8, 32, 16, 16, 128
def timeit(q, k, v, tag):
    for _ in range(100): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
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
    print(f'{tag}: {best:.2f} us', flush=True)
qkv = torch.randn(B, S, H, W, 3 * I, device='cuda').bfloat16()
for rep in range(2):
    timeit(qkv[..., :I], qkv[..., I:2 * I], qkv[..., 2 * I:], 'qkv interleaved [N,384]')
    q = qkv[..., :I].contiguous(); kv = torch.stack([qkv[..., I:2 * I], qkv[..., 2 * I:]]).contiguous()
    timeit(q, kv[0], kv[1], 'q [N,128], k / v planes of one [2,N,128]')
    kvi = qkv[..., I:].contiguous()
    timeit(q, kvi[..., :I], kvi[..., I:], 'q [N,128], k|v interleaved [N,256]')
    pad = torch.empty(2, B, S, H * W + 8, I, device='cuda', dtype=torch.bfloat16)     # planes padded by 8 rows (2 KB)
    pad[:, :, :, :H * W] = kv.view(2, B, S, H * W, I)
    print('(padded planes need a plane stride the C ABI does not have: skipped)')
