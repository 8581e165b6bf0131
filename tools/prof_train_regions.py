"""Which device launches each Python-level region of DenoiserTrainer.forward_backward (fused path) issues: one torch.profiler
session per region, kernel names and counts (the autograd / tensor-op glue around the library's kernels)."""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer, corrupt_last_frame, linear_cross_entropy
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
tr = DenoiserTrainer(m, 1024, distributed=False)
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
r = torch.full((8,), 0.5)
for _ in range(2):
    tr.train_step(z, r=r)
torch.cuda.synchronize()


def region(name, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    ks = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0 and e.device_type.name != 'CPU']
    if not ks:
        ks = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0]
    print(f'--- {name}: {sum(k[1] for k in ks)} device launches, {sum(k[2] for k in ks):.0f} us')
    for k in sorted(ks, key=lambda k: -k[2]):
        print(f'      {k[1]:3d} x {k[0][:90]}  ({k[2]:.0f} us)')
    return out


tr.arena.zero_grad()
zc, target = region('corrupt_last_frame', lambda: corrupt_last_frame(z, r, 1024))
last = region('transformer_forward_train', lambda: fused.transformer_forward_train(m.transformer, zc, last_only=True))
x2 = region('last.reshape', lambda: last.reshape(-1, last.shape[-1]))
mean, rows = region('linear_cross_entropy', lambda: linear_cross_entropy(x2, m.logit_proj.weight, m.logit_proj.bias, target.reshape(-1), chunk=4096))
per = region('per-sample mean', lambda: rows.view(z.shape[0], -1).mean(dim=1))
region('mean.backward()', lambda: mean.backward())
region('optimizer_step', lambda: tr.optimizer_step())
