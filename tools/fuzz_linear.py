"""Randomised shapes / options through the GEMM family (linear forward with LayerNorm prologue, bias, GELU, residual, fp32 output;
data gradient with gelu'; weight gradient with LayerNorm / GELU prologue, bias sums, accumulate / overwrite; the batched weight
gradients incl. the 256-wide kernel) against torch in fp32 (test infrastructure; WMZ_GUARD_ALLOC=1 / 2: over-read detector).

    python3 tools/fuzz_linear.py [cases [seed]]"""
import sys, random, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _guard  # noqa: F401,E402
from world_modelz_amd import ops
F = torch.nn.functional

def rel(a, b, floor=0.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / max(float(b.norm()), floor, 1e-30))

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
bad = 0
for c in range(cases):
    torch.manual_seed(seed * 53 + c)
    M = rng.choice([1, 7, 32, 33, 100, 256, 300, 1000, 2049, 4096, 5000])
    N = 8 * rng.choice([1, 2, 5, 8, 12, 16, 32, 33, 48, 64, 128, 192])
    K = 8 * rng.choice([1, 2, 4, 8, 12, 16, 32, 33, 48, 64, 128])
    dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    kind = rng.choice(['fwd', 'fwd', 'dgrad', 'wgrad', 'wgrad', 'batch'])
    tol = 2e-5 if dt == torch.float32 else 1.5e-2
    a = (torch.randn(M, K) * 0.8 + 0.1).to(dt)
    w = (torch.randn(N, K) / K ** 0.5).to(dt)
    tag = f'case {c}: {kind} M {M} N {N} K {K} {str(dt)[6:]}'
    try:
        if kind == 'fwd':
            use_ln, use_b, use_g, use_r, f32o = (rng.random() < 0.5 for _ in range(5))
            g, b = torch.randn(K) * 0.3 + 1, torch.randn(K) * 0.2
            bias = torch.randn(N)
            res = torch.randn(M, N).to(dt)
            x = a.float()
            if use_ln: x = F.layer_norm(x, (K,), g, b, 1e-5)
            y = x @ w.float().t() + (bias if use_b else 0)
            if use_g: y = F.gelu(y)
            if use_r: y = y + res.float()
            out = ops.linear_fwd(a.cuda(), w.cuda(), bias.cuda() if use_b else None, res.cuda() if use_r else None,
                                 (g.cuda(), b.cuda()) if use_ln else None, gelu=use_g, out_f32=f32o)
            err = rel(out, y)
            tag += f' ln {int(use_ln)} bias {int(use_b)} gelu {int(use_g)} res {int(use_r)} f32out {int(f32o)}'
        elif kind == 'dgrad':
            dc = torch.randn(M, N).to(dt)
            use_z = rng.random() < 0.5
            z = torch.randn(M, K).to(dt)
            ref = dc.float() @ w.float()
            if use_z:
                zz = z.float().requires_grad_(True)
                F.gelu(zz).sum().backward()
                ref = ref * zz.grad
            out = ops.linear_dgrad(dc.cuda(), w.t().contiguous().cuda(), z.cuda() if use_z else None)
            err = rel(out, ref)
            tag += f' gelu\' {int(use_z)}'
        elif kind == 'wgrad':
            dc = (torch.randn(M, N) * 0.3).to(dt)
            use_b, over = rng.random() < 0.5, rng.random() < 0.5
            mode = rng.choice(['plain', 'ln', 'gelu'])
            g, b = torch.randn(K) * 0.3 + 1, torch.randn(K) * 0.2
            x = a.float()
            if mode == 'ln': x = F.layer_norm(x, (K,), g, b, 1e-5)
            if mode == 'gelu': x = F.gelu(x)
            init, binit = torch.randn(N, K), torch.randn(N)
            ref_w = (0 if over else init) + dc.float().t() @ x
            ref_b = (0 if over else binit) + dc.float().sum(0)
            dw, db = init.clone().cuda(), binit.clone().cuda()
            ad = a.cuda()
            stats = ops.layernorm_stats(ad) if mode == 'ln' else None
            ops.linear_wgrad(dc.cuda(), ad, dw, db if use_b else None, ln=(g.cuda(), b.cuda()) if mode == 'ln' else None, ln_stats=stats,
                             gelu_in=mode == 'gelu', overwrite=over)
            err = max(rel(dw, ref_w), rel(db, ref_b) if use_b else 0.0)
            tag += f' {mode} bias {int(use_b)} overwrite {int(over)}'
        else:
            n = rng.choice([2, 3, 5])
            probs, refs = [], []
            for i in range(n):
                Ni, Ki = 8 * rng.choice([12, 16, 24, 32, 48, 64]), 8 * rng.choice([12, 16, 24, 32, 48, 64])
                dc = (torch.randn(M, Ni) * 0.3).to(dt)
                ai = torch.randn(M, Ki).to(dt)
                over, use_b = rng.random() < 0.5, rng.random() < 0.6
                init, binit = torch.randn(Ni, Ki), torch.randn(Ni)
                refs.append(((0 if over else init) + dc.float().t() @ ai.float(), (0 if over else binit) + dc.float().sum(0)))
                probs.append((dc.cuda(), ai.cuda(), init.clone().cuda(), binit.clone().cuda() if use_b else None, over))
            ops.linear_wgrad_batch(probs)
            err = max(max(rel(p[2], r[0]), rel(p[3], r[1]) if p[3] is not None else 0.0) for p, r in zip(probs, refs))
            tag += f' {n} problems'
        torch.cuda.synchronize()
    except Exception as e:                       # noqa: BLE001
        print(f'{tag}: RAISED {type(e).__name__}: {str(e)[:300]}', flush=True)
        bad += 1
        continue
    ok = err < tol
    bad += 0 if ok else 1
    print(f'{tag}: {err:.1e}' + ('' if ok else '   <-- FAIL'), flush=True)
print(f'{bad} bad of {cases}')
sys.exit(1 if bad else 0)
