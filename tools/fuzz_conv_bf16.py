"""Randomised geometries of the conv encoder / decoder in bf16 -- the direct 3x3 / streaming small-K / direct weight-gradient kernels
and the streaming element-wise kernels of round 5 -- forward + every gradient, training-mode BatchNorm, against the library's own
fp32 mode on the same weights (the fp32 mode is the path the oracle tests pin).  Run it under the guard allocator:

    WMZ_GUARD_ALLOC=1 python3 tools/fuzz_conv_bf16.py [cases [seed]]        (=2: tensors at the START of their regions)

every tensor then ends (starts) at the end (start) of its own hipMalloc region, and a kernel that reads or writes past an operand
faults instead of touching a neighbour."""
import sys, random, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402
from world_modelz_amd import config
from world_modelz_amd.train_vqae import VqAutoEncoder


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
bad = ran = 0
for c in range(cases):
    torch.manual_seed(seed * 977 + c)
    E = rng.choice([64, 64, 16, 24])
    hp = rng.choice([128, 128, 64, 32, 40])                   # 128 / 64: the direct kernels; the others fall back
    ds = rng.choice([1, 2, 2, 3])
    Himg, Wimg = rng.choice([(64, 64), (32, 32), (64, 32), (32, 64), (16, 16), (48, 80), (128, 64), (24, 40), (8, 8)])
    B = rng.choice([1, 2, 3, 8])
    ic = rng.choice([3, 3, 1, 4])
    if Himg % (1 << ds) or Wimg % (1 << ds) or B * Himg * Wimg * hp > 6 << 20 or B * (Himg >> ds) * (Wimg >> ds) < 16:
        continue
    m = VqAutoEncoder(embedding_dim=E, num_embeddings=64, downscale_steps=ds, hidden_planes=hp, in_channels=ic).cuda().train()
    x = torch.randn(B, ic, Himg, Wimg, device='cuda')
    zl = torch.randn(B, E, Himg >> ds, Wimg >> ds, device='cuda')
    tag = f'case {c}: E {E} hidden {hp} down {ds} in {ic} img {(Himg, Wimg)} B {B}'
    res = {}
    state = {k: v.clone() for k, v in m.state_dict().items()}
    try:
        for dt in (torch.float32, torch.bfloat16):
            m.load_state_dict(state)
            m.zero_grad(set_to_none=True)
            with config.compute_dtype(dt):
                out = {}
                for name, mod, inp in (('encoder', m.encoder, x), ('decoder', m.decoder, zl)):
                    xd = inp.clone().requires_grad_(True)
                    torch.manual_seed(5)
                    y = mod(xd)
                    w = torch.randn_like(y)
                    (y * w).sum().backward()
                    out[name + '.out'] = y.detach().float()
                    out[name + '.dx'] = xd.grad.float()
                for n, p in m.named_parameters():
                    if p.grad is not None:
                        out['g.' + n] = p.grad.float().clone()
            torch.cuda.synchronize()
            res[dt] = out
        ran += 1
        # (a conv bias in front of a training-mode BatchNorm has a mathematically zero gradient: rounding noise in both modes --
        #  differences are measured against a floor of 1e-2 of the largest parameter gradient)
        floor = 1e-2 * max(float(v.norm()) for k, v in res[torch.float32].items() if k.startswith('g.'))
        worst = max((float((res[torch.bfloat16][k].cpu() - v.cpu()).norm() / max(float(v.norm()), floor if k.startswith('g.') else 1e-30)), k)
                    for k, v in res[torch.float32].items())
        fin = all(torch.isfinite(v).all() for v in res[torch.bfloat16].values())
        # (bf16 activations against fp32: outputs to ~1e-2; a BatchNorm weight or bias gradient -- a bf16 sum over every pixel behind LeakyReLU kinks -- to 0.3-0.45, with the old kernels as with the new)
        ok = fin and rel(res[torch.bfloat16]['encoder.out'], res[torch.float32]['encoder.out']) < 3e-2 and worst[0] < 0.6
        print(f'{tag}: worst {worst[0]:.3f} ({worst[1]}) {"ok" if ok else "FLAGGED"}', flush=True)
        bad += 0 if ok else 1
    except Exception as e:                                    # noqa: BLE001
        print(f'{tag}: EXCEPTION {type(e).__name__}: {e}', flush=True)
        bad += 1
print(f'{bad} flagged of {ran} run')
