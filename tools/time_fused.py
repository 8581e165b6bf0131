"""Per-launch time of the fused per-token kernel at several token counts (graph of 50 launches, one event pair).
FUSED_DBG=1 skips the MFMA loops, =2 the weight DMA + waits (timing ablations only, wmz_debug_fused_knobs)."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
torch.manual_seed(0)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
config.set_compute_dtype(torch.bfloat16)
import os as _os
from world_modelz_amd import _lib as _L
_L.call('wmz_debug_fused_knobs', int(_os.environ.get('FUSED_DBG', '0')))
L = list(m.transformer.layers)
def timeit(fn, reps=50):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / reps)
    return sorted(ts)[1]
import os
XF = int(os.environ.get('XF', '3'))
for planes in [int(p) for p in os.environ.get('PLANES', '1,4,8,16,32,64,128,256,512').split(',')]:
    x = torch.randn(planes, 16, 16, 256, device='cuda').bfloat16()
    o = torch.randn(planes, 16, 16, 128, device='cuda').bfloat16()
    with torch.no_grad():
        ht = timeit(lambda: fused.layer_fused(o, x, L[0], L[1], xflags=XF))
        h = timeit(lambda: fused.layer_fused(o, x, L[0], None, xflags=XF & 1))
        t = timeit(lambda: fused.layer_fused(None, x, None, L[1]))
    if planes % 8 == 0:
        tr = m.transformer
        Bz = planes // 8
        zz = torch.randint(0, 1025, (Bz, 8, 16, 16), device='cuda')
        wp, vc = fused._layer_pack(None, L[0])
        xe = torch.empty(Bz, 8, 16, 16, 256, device='cuda', dtype=torch.bfloat16); qe = torch.empty(Bz, 8, 16, 16, 128, device='cuda', dtype=torch.bfloat16); kve = torch.empty(2, Bz, 8, 16, 16, 128, device='cuda', dtype=torch.bfloat16)
        from world_modelz_amd import _lib as LL
        emb = timeit(lambda: LL.call('wmz_embed_qkv_fused_fwd_planes', zz.data_ptr(), tr.embedding.weight.data_ptr(), tr.pos_emb_s.weight.data_ptr(), tr.pos_emb_h.weight.data_ptr(), tr.pos_emb_w.weight.data_ptr(), xe.data_ptr(), qe.data_ptr(), kve.data_ptr(), wp.data_ptr(), vc.data_ptr(), Bz, 8, 16, 16, 8, 256, 128, 256, 1025, 2, 1e-5, LL.stream()))
        print(f'   embed+tail {emb:7.1f} us')
    print(f'planes {planes:4d} tokens {planes*256:7d} wgs {planes*2:5d}: head+tail {ht:7.1f} us  head {h:7.1f}  tail {t:7.1f}', flush=True)
