"""Randomised model / grid configurations through one training step body (DenoiserTrainer.forward_backward: fused, chain or
op-by-op per-token path, 16-wide / 8-wide / general attention -- whatever the library picks), loss and every parameter gradient
against torch.autograd over the fp32 CPU oracle (test infrastructure: run by hand on the GPU box).

    python3 tools/fuzz_train.py [cases [seed]]"""
import sys, random, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402  (WMZ_GUARD_ALLOC=1: over-read detector)
from world_modelz_amd import config, fused
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
from oracle import train_step as ots

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
bad = 0
for c in range(cases):
    dim, mlp, dh, heads = rng.choice([(256, 256, 128, 1), (256, 256, 128, 1), (96, 256, 128, 1), (384, 512, 128, 1), (64, 96, 32, 2), (128, 128, 64, 2),
                                      (160, 256, 128, 1), (128, 256, 64, 3), (128, 512, 64, 2), (192, 512, 128, 1), (256, 512, 128, 1),
                                      (256, 1024, 128, 2), (512, 1024, 128, 1), (64, 96, 20, 3), (128, 256, 100, 1)])
    H, W = rng.choice([(16, 16), (8, 8), (8, 8), (4, 8), (6, 8), (4, 4), (2, 16), (6, 6), (10, 16)])
    S = rng.choice([1, 2, 3, 4, 6])
    B = rng.choice([1, 2, 3, 4])
    depth = rng.choice([1, 2, 3])
    ext = (rng.choice([1, 2, 3]), rng.choice([0, 1, 3]), rng.choice([0, 1, 3]))
    dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    C = rng.choice([64, 40, 128, 37, 101])
    if B * S * H * W * dim * depth > 3 << 20:
        continue
    torch.manual_seed(seed * 977 + c)
    m = VqVideoDiffusionModel(data_shape=(S, H, W), dim=dim, num_classes=C, extents=ext, depth=depth, dim_head=dh, mlp_dim=mlp, heads=heads)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    z = torch.randint(0, C + 1, (B, S, H, W))
    target = torch.randint(0, C, (B, H, W))
    _, _, loss_ref, grads_ref = ots.step_grads(sd, z, target, ext, heads)
    m = m.cuda()
    tag = f'case {c}: B {B} grid {(S, H, W)} dim {dim} mlp {mlp} {heads}x{dh} depth {depth} ext {ext} C {C} {str(dt)[6:]}'
    try:
        config.set_chain_policy('always')        # (small grids: 'auto' would send every chain-width case to the op-by-op path)
        with config.compute_dtype(dt):
            tr = DenoiserTrainer(m, C, lr=1e-3, warmup=0, max_steps=100, distributed=False)
            path = 'chain' if tr.chain_packs is not None else ('fused' if fused.supported(m.transformer, dt) and z.numel() % 32 == 0 else 'ops')
            tr.arena.zero_grad()
            _, mean = tr.forward_backward(z.cuda(), target.cuda())
            torch.cuda.synchronize()
    except Exception as e:                       # noqa: BLE001
        print(f'{tag}: RAISED {type(e).__name__}: {str(e)[:300]}', flush=True)
        bad += 1
        continue
    floor = 1e-3 * max(float(g.norm()) for g in grads_ref.values())
    errs = {n: float((p.grad.detach().float().cpu() - grads_ref[n]).norm() / max(float(grads_ref[n].norm()), floor)) for n, p in m.named_parameters()}
    worst = max(errs.items(), key=lambda kv: kv[1])
    dl = abs(float(mean) - float(loss_ref))
    tol_g, tol_l = (8e-2, 3e-2) if dt == torch.bfloat16 else (2e-4, 1e-5)
    ok = worst[1] < tol_g and dl < tol_l and all(torch.isfinite(p.grad).all() for p in m.parameters())
    bad += 0 if ok else 1
    print(f'{tag} [{path}]: loss diff {dl:.1e}, worst gradient {worst[1]:.1e} ({worst[0]})' + ('' if ok else '   <-- FAIL'), flush=True)
    del tr, m
print(f'{bad} bad of {cases}')
sys.exit(1 if bad else 0)
