"""The element-wise kernels of the conv path at the frame encoder's shapes (256 frames): affine_act (BatchNorm apply + skip add +
LeakyReLU), BatchNorm backward reduce / apply -- HIP events around 20 launches each, with the bytes they move."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for (H, C, two) in ((64, 64, False), (32, 64, True), (32, 64, False), (16, 64, True), (64, 128, False), (32, 128, False)):
    x = torch.randn(B, H, H, C, device='cuda').bfloat16()
    r = torch.randn(B, H, H, C, device='cuda').bfloat16()
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    mb = x.numel() * 2 / 1e6
    t = timed(lambda: ops.affine_act_nhwc(x, sc, sh, r, sc if two else None, sh if two else None, leaky=True))
    print(f'affine_act  B={B} {H}x{H}x{C} skip{"+affine" if two else ""}: {t:7.1f} us  {3 * mb / t:6.2f} TB/s ({3 * mb:.0f} MB)')
    t = timed(lambda: ops.affine_act_nhwc(x, sc, sh, leaky=True))
    print(f'affine_act  B={B} {H}x{H}x{C} no skip: {t:7.1f} us  {2 * mb / t:6.2f} TB/s ({2 * mb:.0f} MB)')
    mean, rstd = torch.randn(C, device='cuda'), torch.rand(C, device='cuda') + 0.5
    t = timed(lambda: ops.bn_act_bwd(x, r, x, mean, rstd, sc, True))
    print(f'bn_act_bwd  B={B} {H}x{H}x{C} (reduce 3R+1W, apply 2R+1W): {t:7.1f} us  {7 * mb / t:6.2f} TB/s ({7 * mb:.0f} MB)')
    sg = torch.zeros(2, C, device='cuda')
    t = timed(lambda: ops.bn_act_bwd(x, None, r, mean, rstd, sc, True, into=(sg[0], sg[1]), remask=(sc, sh)))
    print(f'bn_leaky_bwd B={B} {H}x{H}x{C} (mask recomputed: reduce 2R, apply 2R+1W): {t:7.1f} us  {5 * mb / t:6.2f} TB/s ({5 * mb:.0f} MB)')
