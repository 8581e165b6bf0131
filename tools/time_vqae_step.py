"""The captured VQ-AE training step (64 frames of 64 x 64, default sizes) with the fused scalar side (quantiser tail + reconstruction
loss kernels) against torch's device ops for the same arithmetic: same box, one process, best of 3 x 20 replays."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
config.set_compute_dtype(torch.bfloat16)
frames = torch.rand(64, 3, 64, 64, device='cuda')
for fused in (True, False, True, False):
    torch.manual_seed(7)
    ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
    tr = VqaeTrainer(ae, distributed=False)
    tr.fused_losses = fused
    tr.enable_graph(frames)
    for _ in range(5):
        tr.train_step(frames)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            tr.train_step(frames)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f'fused losses {fused}: {best * 1e3:.3f} ms per captured step', flush=True)
    del tr, ae
