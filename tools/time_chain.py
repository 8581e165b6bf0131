"""Forward step and training step of the chain kernels' width triples (csrc/chain_widths.h) at config-4 clips (8 x 32 x 16 x 16):
the chain kernels (csrc/layer_chain.hip, layer_chain_bwd.hip) against the per-op path.
    python3 tools/time_chain.py [all | small]  (default: the published widths and the reference's test() geometry; small: config
    5's token count -- 3 072 tokens = 24 workgroups of the chain kernels -- at dim 512 / mlp 1024 / depth 8)"""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.graph import GraphedForward
from world_modelz_amd.train import DenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
config.set_last_frame_cone(False)
SMALL = 'small' in sys.argv[1:]
GRID = (6, 2, 16, 16) if SMALL else (8, 32, 16, 16)
z = torch.randint(0, 1024, GRID, device='cuda')
# (dim, heads, dim_head, mlp, depth, extents)
CASES = [(96, 1, 128, 256, 12, (3, 1, 1)), (384, 1, 128, 512, 20, (3, 1, 1)), (128, 3, 64, 256, 4, (2, 2, 2))]
if SMALL:
    CASES = [(512, 1, 128, 1024, 8, (1, 3, 3)), (256, 2, 128, 1024, 8, (1, 3, 3))]
elif 'all' in sys.argv[1:]:
    CASES += [(128, 2, 64, 512, 4, (3, 3, 3)), (192, 1, 128, 512, 4, (3, 3, 3)), (256, 1, 128, 512, 4, (3, 3, 3)),
              (256, 1, 128, 1024, 4, (3, 3, 3)), (256, 2, 128, 256, 4, (3, 3, 3)), (256, 2, 128, 1024, 4, (3, 3, 3)),
              (512, 1, 128, 1024, 4, (3, 3, 3))]
for dim, heads, dh, mlp, depth, ext in CASES:
    torch.manual_seed(42)
    m = VqVideoDiffusionModel(data_shape=GRID[1:], dim=dim, num_classes=1024, extents=ext, depth=depth, dim_head=dh,
                              mlp_dim=mlp, heads=heads).cuda().eval()
    assert fused.chain_widths(m.transformer) is not None
    outs, line = {}, []
    for mode in ('chain', 'per-op'):
        config.set_chain_policy('always' if mode == 'chain' else 'never')
        with torch.no_grad():
            r = GraphedForward(m, z)
            for _ in range(10): r(r.static_in)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): y = r(r.static_in)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        outs[mode] = y.clone()
        line.append(f'forward {mode} {dt * 1e3:.3f} ms')
        del r
    config.set_chain_policy('always')
    d = (outs['chain'].float() - outs['per-op'].float()).norm() / outs['per-op'].float().norm()
    m.train()
    for mode in ('chain', 'per-op'):
        config.set_fused_training(mode == 'chain')
        try:
            t = DenoiserTrainer(m, 1024, lr=1e-4, warmup=500, max_steps=100000, distributed=False)
            rr = torch.full((GRID[0],), 0.5)
            t.enable_graph(z)
            for _ in range(3): t.train_step(z, r=rr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): t.train_step(z, r=rr)
            torch.cuda.synchronize()
            line.append(f'train {mode} {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms')
            del t
        finally:
            config.set_fused_training(True)
    print(f'dim {dim} {heads}x{dh} mlp {mlp} depth {depth}: ' + ', '.join(line) + f'   (chain vs per-op logits rel {float(d):.2e})', flush=True)
