"""Which ATen ops (copies, fills, reductions) the eager training step still launches around the library's kernels: one step
under torch.profiler, grouped by op and call site (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer, corrupt_last_frame
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
tr = DenoiserTrainer(m, 1024, distributed=False)
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
r = torch.full((8,), 0.5)


def step():
    tr.arena.zero_grad()
    zc, tgt = corrupt_last_frame(z, r, 1024)
    tr.forward_backward(zc, tgt)
    tr.optimizer_step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=50, max_shapes_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=40, max_src_column_width=110))
