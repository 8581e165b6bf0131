import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
torch.manual_seed(0)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
config.set_compute_dtype(torch.bfloat16)
x = torch.randn(8, 32, 16, 16, 256, device='cuda').bfloat16()
o = torch.randn(8, 32, 16, 16, 128, device='cuda').bfloat16()
L = list(m.transformer.layers)
for _ in range(5): fused.layer_fused(o, x, L[0], L[1])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fused.layer_fused(o, x, L[0], L[1])
e1.record(); torch.cuda.synchronize()
print('fused head+tail us:', e0.elapsed_time(e1) * 1000 / 50)
