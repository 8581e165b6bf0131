"""Micro-driver: VqAutoEncoder.encode of 256 64x64 frames, for rocprofv3 (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402  (WMZ_GUARD_ALLOC=1: over-read detector)
from world_modelz_amd.train_vqae import VqAutoEncoder
torch.manual_seed(7)
ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
x = torch.randn(256, 3, 64, 64, device='cuda')
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        t = ae.encode(x)
torch.cuda.synchronize()
