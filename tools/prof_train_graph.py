"""rocprofv3 driver: a captured denoiser training step as hipGraph replays (the last replays of the trace are the steady state).
    python3 tools/prof_train_graph.py config4 | config3 | dim96 | dim384 [replays]"""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
which = sys.argv[1] if len(sys.argv) > 1 else 'config4'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16)
B, S = (16, 16) if which == 'config3' else (8, 32)
dim, mlp, depth, ext = {'config4': (256, 256, 4, (3, 3, 3)), 'config3': (256, 256, 4, (3, 3, 3)), 'dim96': (96, 256, 12, (3, 1, 1)),
                        'dim384': (384, 512, 20, (3, 1, 1))}[which]
m = VqVideoDiffusionModel(data_shape=(S, 16, 16), dim=dim, num_classes=1024, extents=ext, depth=depth, dim_head=128, mlp_dim=mlp,
                          heads=1).cuda()
t = DenoiserTrainer(m, 1024, lr=1e-4, warmup=500, max_steps=200000, distributed=False)
z = torch.randint(0, 1024, (B, S, 16, 16), device='cuda')
r = torch.full((B,), 0.5)
t.enable_graph(z)
for _ in range(n):
    t.train_step(z, r=r)
torch.cuda.synchronize()
