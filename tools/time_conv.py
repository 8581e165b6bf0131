"""Micro-timer for the conv encoder / decoder kernels (not part of the product): every convolution shape of the VQ auto-encoder
(autoencoder.py) at a given frame count, direct / small-K kernels against the implicit-GEMM kernel, HIP events around 20 launches.

    python3 tools/time_conv.py [frames=256]"""
import sys
import torch
sys.path.insert(0, '.')
from world_modelz_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = 'cuda'


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (name, H, W, Cin, Cout, k, stride, pad, kwargs)
shapes = [
    ('E1 conv_1 3x3 8->64 +leaky', 64, 64, 8, 64, 3, 1, 1, dict(leaky=True)),
    ('E2 3x3 64->128 +stats', 64, 64, 64, 128, 3, 1, 1, dict(stats=True)),
    ('E3 1x1 128->64 pre+stats', 64, 64, 128, 64, 1, 1, 0, dict(stats=True, pre=True)),
    ('E4 3x3s2 64->128 +stats', 64, 64, 64, 128, 3, 2, 1, dict(stats=True)),
    ('E5 2x2s2 64->64 +stats', 64, 64, 64, 64, 2, 2, 0, dict(stats=True)),
    ('E2 3x3 64->128 +stats', 32, 32, 64, 128, 3, 1, 1, dict(stats=True)),
    ('E3 1x1 128->64 pre+stats', 32, 32, 128, 64, 1, 1, 0, dict(stats=True, pre=True)),
    ('E4 3x3s2 64->128 +stats', 32, 32, 64, 128, 3, 2, 1, dict(stats=True)),
    ('E5 2x2s2 64->64 +stats', 32, 32, 64, 64, 2, 2, 0, dict(stats=True)),
    ('E3 1x1 128->64 pre+stats', 16, 16, 128, 64, 1, 1, 0, dict(stats=True, pre=True)),
    ('D1 3x3 64->64', 16, 16, 64, 64, 3, 1, 1, dict()),
    ('D2 3x3 64->128 +bias+stats', 32, 32, 64, 128, 3, 1, 1, dict(bias=True, stats=True)),
    ('D3 3x3 128->128 +bias+res', 32, 32, 128, 128, 3, 1, 1, dict(bias=True, res=True)),
    ('D4 1x1 64->128 +bias', 32, 32, 64, 128, 1, 1, 0, dict(bias=True)),
    ('D5 3x3 128->128 +bias+stats', 64, 64, 128, 128, 3, 1, 1, dict(bias=True, stats=True)),
    ('D3 3x3 128->128 +bias+res', 64, 64, 128, 128, 3, 1, 1, dict(bias=True, res=True)),
    ('D4 1x1 128->128 +bias', 64, 64, 128, 128, 1, 1, 0, dict(bias=True)),
    ('D6 3x3 128->8', 64, 64, 128, 8, 3, 1, 1, dict()),
    ('G  3x3 128->64 (dgrad of E2)', 64, 64, 128, 64, 3, 1, 1, dict()),
    ('G  3x3 8->128 (dgrad of D6)', 64, 64, 8, 128, 3, 1, 1, dict()),
]
torch.manual_seed(0)
print(f'frames = {B}')
for name, H, W, Ci, Co, k, s, p, kw in shapes:
    x = torch.randn(B, H, W, Ci, device=dev).bfloat16()
    w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).bfloat16()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    args = dict(stats=kw.get('stats', False), leaky=kw.get('leaky', False))
    if kw.get('bias'):
        args['bias'] = torch.randn(Co, device=dev)
    if kw.get('res'):
        args['residual'] = torch.randn(B, Ho, Wo, Co, device=dev).bfloat16()
    if kw.get('pre'):
        args['pre'] = (torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev), 0.01)
    flops = 2.0 * B * Ho * Wo * Co * k * k * Ci
    byts = 2.0 * (x.numel() + B * Ho * Wo * Co * (2 if kw.get('res') else 1))
    res = []
    for direct in (True, False):
        ops.DIRECT_CONV = direct
        res.append(timed(lambda: ops.conv2d_nhwc(x, w, k, k, s, p, **args)))
    ops.DIRECT_CONV = True
    print(f'{name:32s} {H:3d}x{W:<3d}  new {res[0]:7.1f} us  old {res[1]:7.1f} us   '
          f'{flops / res[0] / 1e6:7.1f} TF/s  {byts / res[0] / 1e6:6.2f} TB/s')
