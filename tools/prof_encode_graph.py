"""rocprofv3 driver: the frozen VQ-AE frame encoder (256 frames of 64 x 64, BatchNorm in training mode) as hipGraph replays."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.graph import GraphedEncoder
from world_modelz_amd.train_vqae import VqAutoEncoder
config.set_compute_dtype(torch.bfloat16)
torch.manual_seed(7)
ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
x = torch.randn(256, 3, 64, 64, device='cuda')
with torch.no_grad():
    enc = GraphedEncoder(ae, x)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        t = enc(x)
torch.cuda.synchronize()
