"""VQ-AE graphed training step with / without the weight-gradient side branch, with / without the skip gradients summed inside
the backward kernels (development timing, one process)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, autoencoder, ops
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
config.set_compute_dtype(torch.bfloat16)
for side, fuse, dil in ((True, True, True), (True, True, False), (True, True, True), (True, True, False), (True, False, True), (False, True, True)):
    config.set_wgrad_stream(side)
    autoencoder.FUSE_SKIP_GRAD = fuse
    ops.DILATE_KERNEL = dil
    torch.manual_seed(7)
    ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
    tr = VqaeTrainer(ae, distributed=False)
    fr = torch.rand(64, 3, 64, 64, device='cuda')
    tr.enable_graph(fr)
    for _ in range(5):
        tr.train_step(fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step(fr)
    torch.cuda.synchronize()
    print(f'side branch {side}, skip gradients fused {fuse}, dilate kernel {dil}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step')
    del tr, ae
