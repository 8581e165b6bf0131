"""Micro-timer of the direct 3x3 weight gradient (csrc/conv_wgrad.hip + the two-stage reduction): 64 frames of 64 x 64 / 32 x 32."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for (B, H, W, Ci, Co) in ((64, 64, 64, 128, 128), (64, 64, 64, 64, 128), (64, 32, 32, 128, 128), (64, 32, 32, 64, 128), (64, 64, 64, 128, 8)):
    x = torch.randn(B, H, W, Ci, device='cuda').bfloat16()
    dy = torch.randn(B, H, W, Co, device='cuda').bfloat16()
    t = timed(lambda: ops.conv2d_nhwc_wgrad(x, dy, 3, 3, 1, 1, True))
    fl = 2.0 * B * H * W * Co * 9 * Ci
    print(f'wgrad 3x3 {Ci:3d}->{Co:3d} {B} x {H}x{W}: {t:7.1f} us (kernel + reduce)  {fl / t / 1e6:7.1f} TF/s', flush=True)
