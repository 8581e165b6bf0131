import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
from oracle import denoiser as oden
def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
for (S, extents, depth, B) in [(7, (2, 1, 1), 1, 2), (7, (2, 1, 1), 2, 2), (7, (3, 3, 3), 1, 2), (7, (2, 3, 3), 1, 2), (7, (2, 1, 3), 1, 2), (7, (2, 3, 1), 1, 2)]:
    torch.manual_seed(11)
    m = VqVideoDiffusionModel(data_shape=(S, 16, 16), dim=256, num_classes=257, extents=extents, depth=depth, dim_head=128, mlp_dim=256, heads=1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'norm' in n or n.endswith('bias'):
                p.add_(0.2 * torch.randn_like(p))
    m = m.cuda().eval()
    z = torch.randint(0, 258, (B, S, 16, 16)).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref = oden.denoiser_forward(sd, z.cpu(), extents, 1)
    config.set_last_frame_cone(False)
    with torch.no_grad():
        config.set_compute_dtype(torch.bfloat16)
        full = m(z)
    perop = m(z)
    config.set_compute_dtype(torch.float32)
    f32 = m(z)
    print(S, extents, depth, 'fused', rel(full, ref), 'perop bf16', rel(perop, ref), 'fp32', rel(f32, ref), 'ref norm', float(ref.norm()))
