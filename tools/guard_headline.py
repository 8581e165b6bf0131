"""The inference paths of bench.py as eager launches (full-grid forward, last-frame cone, published widths on the chain kernel,
reference geometry with 8-wide planes, one sampler iteration) for a WMZ_GUARD_ALLOC=1 / 2 sweep (tools/_guard.py)."""
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
config.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
def model(shape, dim, mlp, depth, ext, C=1024):
    return VqVideoDiffusionModel(data_shape=shape, dim=dim, num_classes=C, extents=ext, depth=depth, dim_head=128, mlp_dim=mlp, heads=1).cuda().eval()
for name, shape, B, dim, mlp, depth, ext in (('config 4', (32, 16, 16), 8, 256, 256, 4, (3, 3, 3)), ('dim 384', (32, 16, 16), 4, 384, 512, 4, (3, 1, 1)),
                                             ('dim 96', (32, 16, 16), 4, 96, 256, 3, (3, 1, 1)), ('reference geometry', (6, 8, 8), 64, 384, 512, 4, (3, 1, 1)),
                                             ('ragged', (5, 6, 8), 3, 256, 256, 2, (2, 1, 1)), ('one row', (3, 1, 16), 2, 256, 256, 2, (1, 0, 2))):
    m = model(shape, dim, mlp, depth, ext)
    z = torch.randint(0, 1025, (B,) + shape, device='cuda')
    with torch.no_grad():
        for cone in (False, True):
            config.set_last_frame_cone(cone)
            y = m(z)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all()
    print(name, 'ok', tuple(y.shape), flush=True)
    del m
print('ALL OK')
