"""Forward step of the reference's published widths (dim 96 / mlp 256 / depth 12, dim 384 / mlp 512 / depth 20, window 7x3x3) at
config-4 clips: the chain kernel (csrc/layer_chain.hip) against the per-op path."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.graph import GraphedForward
config.set_compute_dtype(torch.bfloat16)
config.set_last_frame_cone(False)
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
orig = fused.chain_supported
for dim, mlp, depth in ((96, 256, 12), (384, 512, 20)):
    torch.manual_seed(42)
    m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=dim, num_classes=1024, extents=(3, 1, 1), depth=depth, dim_head=128,
                              mlp_dim=mlp, heads=1).cuda().eval()
    outs = {}
    for mode in ('chain', 'per-op'):
        fused.chain_supported = orig if mode == 'chain' else (lambda *a: False)
        with torch.no_grad():
            r = GraphedForward(m, z)
            for _ in range(10): r(r.static_in)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): y = r(r.static_in)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        outs[mode] = y.clone()
        print(f'dim {dim} depth {depth} {mode}: {dt * 1e3:.3f} ms/step ({dt / depth * 1e6:.1f} us per layer)', flush=True)
    d = (outs['chain'].float() - outs['per-op'].float()).norm() / outs['per-op'].float().norm()
    print(f'   chain vs per-op logits: rel {float(d):.2e}')
fused.chain_supported = orig
