"""Micro-driver: a few config-5 (sparse) training steps for rocprofv3 (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402  (WMZ_GUARD_ALLOC=1: over-read detector)
from world_modelz_amd import config
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
from world_modelz_amd.train import SparseDenoiserTrainer
torch.manual_seed(43)
config.set_compute_dtype(torch.bfloat16)
sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
zs = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
rs = torch.full((6,), 0.5)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    st.train_step(zs, r=rs)
torch.cuda.synchronize()
