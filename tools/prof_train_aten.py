"""Which Python lines issue the ATen glue launches (copy_, fill_, zero_, elementwise) of one eager training step at config 4."""
import sys, collections, torch
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
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::mul', 'aten::sum', 'aten::mean', 'aten::div', 'aten::add_',
                  'aten::mul_', 'aten::sub', 'aten::_to_copy', 'aten::cat', 'aten::index_select', 'aten::gather', 'aten::clone'):
        st = [s for s in (e.stack or []) if 'world_modelz' in s or 'world-modelz' in s or 'prof_train_aten' in s]
        shp = str(getattr(e, 'input_shapes', ''))
        cnt[(e.name, shp)] += 1
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(c, n, s)
print('--- in order')
for e in prof.events():
    if e.name.startswith('aten::') and e.name not in ('aten::empty', 'aten::view', 'aten::reshape', 'aten::empty_strided', 'aten::as_strided', 'aten::select', 'aten::slice', 'aten::detach', 'aten::_unsafe_view', 'aten::empty_like', 'aten::t', 'aten::transpose', 'aten::expand', 'aten::alias', 'aten::unsqueeze', 'aten::squeeze', 'aten::contiguous', 'aten::to', 'aten::item', 'aten::_local_scalar_dense', 'aten::lift_fresh', 'aten::zeros', 'aten::ones', 'aten::full', 'aten::result_type', 'aten::view_as', 'aten::narrow', 'aten::unbind', 'aten::resolve_conj', 'aten::resolve_neg', 'aten::is_nonzero'):
        print(e.name, getattr(e, 'input_shapes', ''))
