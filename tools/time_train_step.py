"""The captured training steps (config 4 denoiser, config 3 clips, config 5 sparse) with the linear + cross-entropy's parameter
gradients (a) through autograd (zero fills, scalings, accumulations: what ran until round 6 -- the arena shortcut was dead code),
(b) straight into the arena on the compute stream, (c) straight into the arena on the weight-gradient side branch.  Same box, one
process, best of 3 x 20 replays."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, train
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
config.set_compute_dtype(torch.bfloat16)


def timed(tr, z, r):
    tr.enable_graph(z)
    for _ in range(5):
        tr.train_step(z, r=r)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            tr.train_step(z, r=r)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    return best * 1e3


for name in ('config 4 (8 x 32x16x16)', 'config 3 (16 x 16x16x16)', 'config 5 (sparse)'):
    for rep in range(2):
        line = []
        for mode, (d, s_) in (('autograd', (False, False)), ('arena', (True, False)), ('arena + side branch', (True, True))):
            train.CE_DIRECT, train.CE_SIDE = d, s_
            torch.manual_seed(42)
            if 'sparse' in name:
                m = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
                tr = train.SparseDenoiserTrainer(m, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
                z = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
            else:
                B, S = (8, 32) if 'config 4' in name else (16, 16)
                m = VqVideoDiffusionModel(data_shape=(S, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128,
                                          mlp_dim=256, heads=1).cuda()
                tr = train.DenoiserTrainer(m, 1024, lr=1e-4, warmup=500, max_steps=200000, distributed=False)
                z = torch.randint(0, 1024, (B, S, 16, 16), device='cuda')
            line.append(f'{mode} {timed(tr, z, torch.full((z.shape[0],), 0.5)):.3f}')
            del tr, m
        print(f'{name}: ' + ', '.join(line) + ' ms', flush=True)
train.CE_DIRECT, train.CE_SIDE = True, None
