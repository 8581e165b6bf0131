import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
qkv = torch.randn(8, 32, 16, 16, 384, device='cuda').bfloat16()
q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
for _ in range(5): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
e1.record(); torch.cuda.synchronize()
print('attn fwd us:', e0.elapsed_time(e1) * 1000 / 50)
