"""Where the chain kernels (128-token workgroups) overtake the per-op path as the token count grows: forward and training step of
three width triples at 3 072 .. 32 768 tokens (fused.CHAIN_MIN_TOKENS is read off this table)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.graph import GraphedForward
from world_modelz_amd.train import DenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
config.set_last_frame_cone(False)
for dim, heads, dh, mlp, depth in ((96, 1, 128, 256, 4), (128, 3, 64, 256, 4), (384, 1, 128, 512, 4), (512, 1, 128, 1024, 4)):
    for grid in ((6, 2, 16, 16), (6, 4, 16, 16), (8, 4, 16, 16), (6, 8, 16, 16), (8, 8, 16, 16), (8, 16, 16, 16)):
        torch.manual_seed(42)
        z = torch.randint(0, 1024, grid, device='cuda')
        m = VqVideoDiffusionModel(data_shape=grid[1:], dim=dim, num_classes=1024, extents=(1, 3, 3), depth=depth, dim_head=dh,
                                  mlp_dim=mlp, heads=heads).cuda().eval()
        line = []
        for mode in ('chain', 'per-op'):
            config.set_chain_policy('always' if mode == 'chain' else 'never')
            with torch.no_grad():
                r = GraphedForward(m, z)
                for _ in range(10): r(r.static_in)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30): r(r.static_in)
                torch.cuda.synchronize()
                line.append(f'fwd {mode} {(time.perf_counter() - t0) / 30 * 1e3:.3f}')
            del r
        config.set_chain_policy('always')
        m.train()
        for mode in ('chain', 'per-op'):
            config.set_fused_training(mode == 'chain')
            try:
                t = DenoiserTrainer(m, 1024, lr=1e-4, warmup=500, max_steps=100000, distributed=False)
                rr = torch.full((grid[0],), 0.5)
                t.enable_graph(z)
                for _ in range(3): t.train_step(z, r=rr)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8): t.train_step(z, r=rr)
                torch.cuda.synchronize()
                line.append(f'train {mode} {(time.perf_counter() - t0) / 8 * 1e3:.2f}')
                del t
            finally:
                config.set_fused_training(True)
        n = grid[0] * grid[1] * grid[2] * grid[3]
        print(f'dim {dim} mlp {mlp} x{depth}, {n:6d} tokens ({n // 128:3d} workgroups): ' + ', '.join(line) + ' ms', flush=True)
