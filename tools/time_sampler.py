"""Time of one denoise iteration of the sampler loop (sample.py) at config 4: forward graph vs the sampling glue around it."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.sample import sample_frames
torch.manual_seed(0)
config.set_compute_dtype(torch.bfloat16)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda().eval()
z = torch.randint(0, 1024, (8, 32, 16, 16), device='cuda')
for topk in (-1, 100):
    sample_frames(m, z, 1024, 1, num_eval_iterations=30, sample_topk=topk)      # captures the step (kept with the model)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sample_frames(m, z, 1024, 2, num_eval_iterations=30, sample_topk=topk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 60
    print(f'top-k {topk}: {dt * 1e3:.3f} ms per denoise iteration (8 clips; forward alone ~0.30 ms)')
