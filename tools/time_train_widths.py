"""Training step of the reference's published widths (dim 96 / mlp 256 / depth 12, dim 384 / mlp 512 / depth 20, window 7x3x3) at
config-4 clips, graph replay and eager, plus the default widths for comparison."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, ops
import os
if os.environ.get('KEEP_NORM') == '0':
    ops.KEEP_NORM_MIN_ROWS = 1 << 30     # A/B: weight gradients re-normalise their operand (no LN(x) kept, 128-wide tiles)
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
z = torch.randint(0, 1024, (8, 32, 16, 16), device='cuda')
r = torch.full((8,), 0.5)
for dim, mlp, depth, ext in ((256, 256, 4, (2, 2, 2)), (96, 256, 12, (3, 1, 1)), (384, 512, 20, (3, 1, 1))):
    torch.manual_seed(42)
    m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=dim, num_classes=1024, extents=ext, depth=depth, dim_head=128,
                              mlp_dim=mlp, heads=1).cuda().train()
    tr = DenoiserTrainer(m, 1024, lr=1e-4, warmup=500, max_steps=200000)
    for mode in ('eager', 'graph'):
        if mode == 'graph':
            tr.enable_graph(z)
        for _ in range(3): tr.train_step(z, r=r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n): tr.train_step(z, r=r)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'dim {dim} mlp {mlp} depth {depth} {mode}: {dt * 1e3:.2f} ms/step ({dt / depth * 1e6:.0f} us per layer)', flush=True)
    del tr, m
