"""Which ATen ops the VQ-AE training step launches around the library's kernels: one eager step under torch.profiler."""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from world_modelz_amd import config
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
torch.manual_seed(7)
config.set_compute_dtype(torch.bfloat16)
ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
tr = VqaeTrainer(ae, distributed=False)
frames = torch.rand(64, 3, 64, 64, device='cuda')
for _ in range(3):
    tr.train_step(frames)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(frames)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cuda_time_total', row_limit=50, max_name_column_width=40, max_src_column_width=120))
