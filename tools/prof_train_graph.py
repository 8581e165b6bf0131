"""rocprofv3 driver: the config-4 training step as hipGraph replays (the last replay of the trace = the steady state).
prof_train_graph.py [replays [dim mlp depth eS eH eW]]"""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16)
dim, mlp, depth, eS, eH, eW = (int(v) for v in sys.argv[2:8]) if len(sys.argv) >= 8 else (256, 256, 4, 3, 3, 3)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=dim, num_classes=1024, extents=(eS, eH, eW), depth=depth, dim_head=128, mlp_dim=mlp, heads=1).cuda()
tr = DenoiserTrainer(m, 1024, distributed=False)
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
r = torch.full((8,), 0.5)
tr.enable_graph(z)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    tr.train_step(z, r=r)
torch.cuda.synchronize()
