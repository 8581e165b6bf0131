"""Micro-driver: a few training steps at config 4 for rocprofv3 (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer, corrupt_last_frame
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
tr = DenoiserTrainer(m, 1024, distributed=False)
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
r = torch.full((8,), 0.5)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    tr.arena.zero_grad()
    zc, tgt = corrupt_last_frame(z, r, 1024)
    tr.forward_backward(zc, tgt)
    tr.optimizer_step()
torch.cuda.synchronize()
