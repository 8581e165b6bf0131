"""Config 5's sampler (sparse_diffusion.sample_clips: the token side of the reference's evaluate_model) at full size -- 8 clips of
64 x 16 x 16, codebook 8192, 512-token contexts: time per denoise sub-step (context draw + forward + multinomial + scatter)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel, sample_clips
config.set_compute_dtype(torch.bfloat16)
torch.manual_seed(1)
m = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
for mode in ('neighbors', 'uniform'):
    sample_clips(m, 8, 8192, mode, 512, 2, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 4
    z = sample_clips(m, 8, 8192, mode, 512, it, seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sub = it * (64 * 256 // 512 + 1)
    print(f'{mode}: {dt / sub * 1e3:.3f} ms per sub-step ({sub} sub-steps, {dt:.2f} s; the reference runs 100 iterations x 33), '
          f'masked left {float((z == 8192).float().mean()):.3f}', flush=True)
