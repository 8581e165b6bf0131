"""Frame encoder (graphed, 256 frames) and VQ-AE graphed training step with every BatchNorm finalised by the kernel that applies it
(ops.BnLazy, include/wmz.h wmz_bn_stats) against a wmz_bn_finalize launch between each convolution and its consumer, alternating in
one process (development timing)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, ops
from world_modelz_amd.graph import GraphedEncoder
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
config.set_compute_dtype(torch.bfloat16)


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    for lazy in (True, False):
        ops.BN_LAZY = lazy
        torch.manual_seed(7)
        ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
        frames = torch.rand(256, 3, 64, 64, device='cuda')
        enc = GraphedEncoder(ae, frames)
        t_enc = timed(lambda: enc(frames), 50)
        tr = VqaeTrainer(ae, distributed=False)
        fr = torch.rand(64, 3, 64, 64, device='cuda')
        tr.enable_graph(fr)
        t_tr = timed(lambda: tr.train_step(fr), 20)
        print(f'BatchNorm finalised by its consumer {lazy}: encoder {t_enc:.3f} ms per 256 frames, VQ-AE step {t_tr:.3f} ms', flush=True)
        del tr, enc, ae
ops.BN_LAZY = True
