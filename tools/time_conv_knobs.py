"""Ablation / skew sweep of the direct 3x3 convolution (development; not part of the product)."""
import sys
import torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L  # noqa: E402
B = 256


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


for (H, W, Ci, Co, st) in ((64, 64, 64, 128, True), (64, 64, 128, 128, True), (32, 32, 64, 128, True)):
    x = torch.randn(B, H, W, Ci, device='cuda').bfloat16()
    w = (torch.randn(Co, 9 * Ci, device='cuda') * 0.05).bfloat16()
    for skew, dbg in ((1, 0), (1, 32), (1, 64), (1, 1)):          # product | no stores | stores only | no epilogue
        L.call('wmz_debug_conv_knobs', skew, dbg)
        t = timed(lambda: ops.conv2d_nhwc(x, w, 3, 3, 1, 1, stats=st))
        print(f'{H}x{W} {Ci}->{Co} skew {skew} dbg {dbg}: {t:7.1f} us')
    L.call('wmz_debug_conv_knobs', 0, 0)
