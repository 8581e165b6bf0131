"""Micro-driver: the config-4 forward step in the precise (IEEE-half) fused mode, eager launches, for rocprofv3 (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
mode = sys.argv[2] if len(sys.argv) > 2 else 'f16'
config.set_compute_dtype({'f16': torch.float16, 'bf16': torch.bfloat16}[mode])
config.set_last_frame_cone(False)
config.set_clip_streams(1)
torch.manual_seed(42)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda().eval()
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        y = m(z)
torch.cuda.synchronize()
