"""Micro-driver: graphed VQ-AE training steps for rocprofv3 (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
torch.manual_seed(7)
config.set_compute_dtype(torch.bfloat16)
ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
tr = VqaeTrainer(ae, distributed=False)
frames = torch.rand(64, 3, 64, 64, device='cuda')
tr.enable_graph(frames)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    tr.train_step(frames)
torch.cuda.synchronize()
