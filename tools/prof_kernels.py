"""Micro-driver for profiling single kernels under rocprofv3 (not part of the product).

    python3 tools/prof_kernels.py {fused|attn|attn_bwd|vq} N        config-4 shapes (65 536 tokens, dh 128, window 7x7x7)
"""
import sys
import torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused, ops
from world_modelz_amd.main import VqVideoDiffusionModel

which = sys.argv[1] if len(sys.argv) > 1 else 'fused'
torch.manual_seed(0)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128,
                          mlp_dim=256, heads=1).cuda()
config.set_compute_dtype(torch.bfloat16)
N = 65536
x = torch.randn(8, 32, 16, 16, 256, device='cuda').bfloat16()
o = torch.randn(8, 32, 16, 16, 128, device='cuda').bfloat16()
layers = list(m.transformer.layers)
# the layouts of the training step: q its own [N, I] buffer, k | v the two column halves of one [N, 2I] buffer
q = torch.randn(8, 32, 16, 16, 128, device='cuda').bfloat16()
kv = torch.randn(8, 32, 16, 16, 256, device='cuda').bfloat16()
k, v = kv[..., :128], kv[..., 128:]
if which == 'attn_bwd':
    out, lse, _ = ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1, need_lse=True)
    dout = torch.randn_like(out)
if which == 'vq':
    xq = torch.randn(65536, 64, device='cuda')
    cb = torch.randn(1024, 64, device='cuda')
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    if which == 'fused':
        fused.layer_fused(o, x, layers[0], layers[1])
    elif which == 'attn':
        ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
    elif which == 'attn_bwd':
        ops.local3d_attention_bwd(q, k, v, out, lse, dout, (3, 3, 3), 1)
    elif which == 'vq':
        ops.vq_argmin(xq, cb)
torch.cuda.synchronize()
