"""Attention forward launch time on the denoiser's own layer-0 q / k / v (random-init weights, random tokens: bench.py's
inputs) beside random-normal q / k / v of the same shape, and the spread of the scaled logits in both."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, fused, config, functional as Fw
from world_modelz_amd.main import VqVideoDiffusionModel
torch.manual_seed(42)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128,
                          mlp_dim=256, heads=1).cuda()
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
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
    s = (q[0, 5, 8].float() @ k[0, 5, 8].float().t()) * 128 ** -0.5
    print(f'{tag}: {best:.2f} us   (logit std {float(s.std()):.2f}, max {float(s.abs().max()):.2f}; |q| {float(q.float().norm(dim=-1).mean()):.2f} |k| {float(k.float().norm(dim=-1).mean()):.2f})', flush=True)
with torch.no_grad(), config.compute_dtype(torch.bfloat16):
    tr = m.transformer
    x = Fw.embed_tokens(z, tr.embedding.weight, tr.pos_emb_s.weight, tr.pos_emb_h.weight, tr.pos_emb_w.weight)
    _, q, kv = fused.layer_fused(None, x, None, tr.layers[0])
    r = torch.randn(3, 8, 32, 16, 16, 128, device='cuda').bfloat16()
    for rep in range(2):
        timeit(r[0], r[1], r[2], 'randn q k v        ')
        timeit(q, kv[0], kv[1], 'model layer-0 q k v')
        timeit(q, kv[0], r[2], 'model q k, randn v ')
        timeit(r[0], r[1], kv[1], 'randn q k, model v ')
