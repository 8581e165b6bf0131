"""Micro-driver: a few training steps at config 4 for rocprofv3 (not part of the product).
prof_train.py [steps [dim mlp depth eS eH eW]] -- e.g. `8 384 512 20 3 1 1` = the reference's published dim-384 run."""
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402  (WMZ_GUARD_ALLOC=1: over-read detector)
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer, corrupt_last_frame
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16)
dim, mlp, depth, eS, eH, eW = (int(v) for v in sys.argv[2:8]) if len(sys.argv) >= 8 else (256, 256, 4, 3, 3, 3)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=dim, num_classes=1024, extents=(eS, eH, eW), depth=depth, dim_head=128, mlp_dim=mlp, heads=1).cuda()
tr = DenoiserTrainer(m, 1024, distributed=False)
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
r = torch.full((8,), 0.5)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    tr.arena.zero_grad()
    zc, tgt = corrupt_last_frame(z, r, 1024)
    tr.forward_backward(zc, tgt)
    tr.optimizer_step()
torch.cuda.synchronize()
