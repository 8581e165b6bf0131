"""Which ATen ops (copies, fills, ...) the config-5 training step launches around the library's kernels: one eager step under
torch.profiler, grouped by op + input shapes and by call site (not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from world_modelz_amd import config
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
from world_modelz_amd.train import SparseDenoiserTrainer
torch.manual_seed(43)
config.set_compute_dtype(torch.bfloat16)
sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
zs = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
rs = torch.full((6,), 0.5)
for _ in range(3):
    st.train_step(zs, r=rs)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    st.train_step(zs, r=rs)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=50, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=5).table(sort_by='self_cuda_time_total', row_limit=60, max_name_column_width=40, max_src_column_width=120))
