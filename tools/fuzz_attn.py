"""Randomised geometries through the attention core, forward and backward, against the fp32 CPU oracle's autograd (test
infrastructure: run by hand on the GPU box; a failing case is added to tests/ as a named case).

    python3 tools/fuzz_attn.py [cases [seed]]

Per case: random (B, S, H, W), heads x dim_head, extents (including windows wider than the grid and extent 0), bf16 or fp32; the
library picks its own path (16-wide / 8-wide row kernels, small-plane shape, general kernel).  Checked: out, lse, dq | dk | dv."""
import sys, random, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402  (WMZ_GUARD_ALLOC=1: over-read detector)
from world_modelz_amd import ops
from oracle import attention as oat

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
bad = 0
for c in range(cases):
    W = rng.choice([16, 16, 8, 8, 4, 5, 12, 32])
    H = rng.choice([16, 8, 6, 4, 2, 10, 1, 12, 32]) if W in (16, 8) else rng.choice([3, 4, 7, 16])
    S = rng.choice([1, 2, 3, 5, 8])
    B = rng.choice([1, 2, 3])
    heads = rng.choice([1, 1, 2, 4])
    dh = rng.choice([128, 128, 64, 32, 16, 8])
    ext = (rng.choice([0, 1, 2, 3, 9]), rng.choice([0, 1, 2, 3, 9]), rng.choice([0, 1, 2, 3, 9]))
    dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    if B * S * H * W * heads * dh > 1 << 18:
        continue
    torch.manual_seed(seed * 1000 + c)
    I = heads * dh
    q, k, v = (torch.randn(B, S, H, W, I).to(dt) for _ in range(3))
    qo, ko, vo = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = oat.local_attention(ko, vo, qo, ext, heads)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    try:
        out, lse, _ = ops.local3d_attention_fwd(qd, kd, vd, ext, heads, need_lse=True)
        dq, dkv = ops.local3d_attention_bwd(qd, kd, vd, out, lse, w.to(dt).cuda(), ext, heads)
        torch.cuda.synchronize()
    except Exception as e:                       # noqa: BLE001
        print(f'case {c}: {(B, S, H, W)} heads {heads} dh {dh} ext {ext} {dt}: RAISED {type(e).__name__}: {str(e)[:200]}', flush=True)
        bad += 1
        continue
    tol = 2e-5 if dt == torch.float32 else 1.2e-2
    # the backward's reference gradient is for the fp32 w; the library saw w rounded to dt: compare against that
    qo.grad = ko.grad = vo.grad = None
    (oat.local_attention(ko, vo, qo, ext, heads) * w.to(dt).float()).sum().backward()
    dk, dv = dkv[..., :I], dkv[..., I:]
    # (a window of one key has exactly-zero dq / dk in exact arithmetic: measure those against the gradients' common scale)
    floor = 1e-3 * max(float(t.norm()) for t in (qo.grad, ko.grad, vo.grad))
    relf = lambda a, b: float((a.detach().float().cpu() - b).norm() / max(float(b.norm()), floor))
    errs = dict(out=rel(out, ref), dq=relf(dq, qo.grad), dk=relf(dk, ko.grad), dv=relf(dv, vo.grad))
    lse_ref = oat.local_attention_lse(q.float(), k.float(), ext, heads).reshape(-1, heads)
    errs['lse'] = float((lse.cpu().reshape(-1, heads) - lse_ref).abs().max() / (lse_ref.abs().max() + 1e-30))
    ok = all(e < tol for e in errs.values()) and all(torch.isfinite(t).all() for t in (out, dq, dkv))
    if not ok:
        bad += 1
    print(f'case {c}: {(B, S, H, W)} heads {heads} dh {dh} ext {ext} {str(dt)[6:]}: ' + ' '.join(f'{n} {e:.1e}' for n, e in errs.items())
          + ('' if ok else '   <-- FAIL'), flush=True)
print(f'{bad} bad of {cases}')
sys.exit(1 if bad else 0)
