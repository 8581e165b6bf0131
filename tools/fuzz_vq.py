"""Randomised (N, C, E, scale, duplicate / near-tie structure) through the VQ nearest-code search against the CPU oracle's pinned
fp32 formula (oracle.vq.distances_avx_order = the order ATen's CPU kernel sums in: indices AND minimum distances must be equal
bit for bit) and the screened search against the exact scan (test infrastructure: run by hand on the GPU box).

    python3 tools/fuzz_vq.py [cases [seed]]"""
import sys, random, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402  (WMZ_GUARD_ALLOC=1: over-read detector)
from world_modelz_amd import ops
from oracle import vq as ovq
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
bad = 0
for c in range(cases):
    E = rng.choice([64, 64, 64, 16, 8, 32, 128, 24, 200, 256, 504])
    C = rng.choice([512, 1024, 64, 128, 256, 40, 1000, 2048, 8192])
    N = rng.choice([1, 7, 64, 1000, 4096, 5000, 20000])
    scale = rng.choice([1.0, 1.0, 1e-3, 50.0])
    if N * C * E > 1 << 25:                     # (the oracle's pinned-order formula materialises [N, C, E])
        N = max(1, (1 << 25) // (C * E))
    g = torch.Generator().manual_seed(seed * 7919 + c)
    x, cb = torch.randn(N, E, generator=g) * scale, torch.randn(C, E, generator=g) * scale
    kind = rng.choice(['plain', 'dups', 'cluster', 'on_codes'])
    if kind == 'dups' and C > 8:
        cb[C // 2] = cb[3]; cb[C - 1] = cb[3]
        x[: min(N, 16)] = cb[3] + 1e-4 * scale * torch.randn(min(N, 16), E, generator=g)
    elif kind == 'cluster':
        k = max(1, C // 32)
        cb = (torch.randn(k, E, generator=g) * scale).repeat_interleave((C + k - 1) // k, 0)[:C] + 1e-4 * scale * torch.randn(C, E, generator=g)
    elif kind == 'on_codes':
        n = min(N, C)
        x[:n] = cb[:n]
    xd, cbd = x.cuda(), cb.cuda()
    idx, d = ops.vq_argmin(xd, cbd, need_dist=True)
    idx_e, d_e = ops.vq_argmin(xd, cbd, need_dist=True, exact_scan=True)
    dref = ovq.distances_avx_order(x, cb[None])[:, 0]
    dmin_ref, idx_ref = dref.min(dim=1)
    ok = torch.equal(idx, idx_e) and torch.equal(d, d_e) and torch.equal(idx.cpu(), idx_ref) and torch.equal(d.cpu(), dmin_ref)
    bad += 0 if ok else 1
    print(f'case {c}: N {N} C {C} E {E} scale {scale} {kind}: ' + ('ok' if ok else
          f'MISMATCH idx vs exact {int((idx != idx_e).sum())}, vs oracle {int((idx.cpu() != idx_ref).sum())}, dist vs oracle {int((d.cpu() != dmin_ref).sum())}   <-- FAIL'), flush=True)
print(f'{bad} bad of {cases}')
sys.exit(1 if bad else 0)
