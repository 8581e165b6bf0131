"""GPU parity: each HIP kernel, called through the C ABI (ctypes), against the CPU oracle and the golden
vectors captured from the reference.  Tolerances are written next to each check."""
import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu

from oracle import attention as oat          # noqa: E402
from oracle import denoiser as oden          # noqa: E402
from oracle import vq as ovq                 # noqa: E402


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a ROCm device'
    from world_modelz_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---------------------------------------------------------------------------------------------- attention

@pytest.mark.parametrize('tag', list('abcdef'))
def test_attention_fp32_vs_golden(ops, tag):
    """fp32 mode (exact-f32 MFMA): out and logits against the reference capture.  1e-5 rel on out;
    logits at live slots 1e-5 rel / 1e-5 abs; masked slots are the literal -1e9."""
    g = load_golden(f'attn_core_{tag}')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    out, lse, dbg = ops.local3d_attention_fwd(dev(g['q']), dev(g['k']), dev(g['v']), ext, heads, need_lse=True,
                                              logits_dbg=True)
    assert rel(out, g['out']) < 1e-5
    B, S, H, W, I = g['q'].shape
    logits = dbg.cpu().reshape(B, S, H, W, heads, -1)
    assert torch.equal(logits == -1e9, g['logits'] == -1e9)
    live = g['logits'] != -1e9
    assert torch.allclose(logits[live], g['logits'][live], rtol=1e-5, atol=1e-5)
    lse_ref = torch.logsumexp(g['logits'], dim=-1).reshape(-1, heads)
    assert torch.allclose(lse.cpu(), lse_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tag', ['e', 'f'])
def test_attention_bf16_logits_vs_golden(ops, tag):
    """bf16 mode on bf16-representable q,k,v (SURVEY 7 'Parity definition' (i)): logits <= 1e-3 rel
    (north_star bar; measured ~1e-6 because bf16 products are exact in the fp32 accumulator);
    out within 1e-2 rel (P is rounded to bf16 for the PV MFMA)."""
    g = load_golden(f'attn_core_{tag}')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    q, k, v = (dev(g[n]).to(torch.bfloat16) for n in 'qkv')
    assert torch.equal(q.float().cpu(), g['q'])          # inputs are exactly representable
    out, _, dbg = ops.local3d_attention_fwd(q, k, v, ext, heads, logits_dbg=True)
    B, S, H, W, I = g['q'].shape
    logits = dbg.cpu().reshape(B, S, H, W, heads, -1)
    assert torch.equal(logits == -1e9, g['logits'] == -1e9)
    live = g['logits'] != -1e9
    err = (logits[live] - g['logits'][live]).abs().max() / g['logits'][live].abs().max()
    assert float(err) < 1e-3
    assert rel(logits[live], g['logits'][live]) < 1e-3
    assert rel(out, g['out']) < 1e-2


@pytest.mark.parametrize('shape,heads,dh,ext,dtype', [
    ((1, 8, 8, 8), 1, 128, (3, 3, 3), torch.bfloat16),     # BASELINE config 2
    ((1, 8, 8, 8), 1, 128, (3, 1, 1), torch.bfloat16),
    ((2, 5, 16, 16), 1, 128, (3, 3, 3), torch.bfloat16),   # 16x16 planes as in configs 3/4
    ((1, 3, 6, 40), 2, 64, (1, 2, 3), torch.bfloat16),     # W > 16: tiles narrower than a row
    ((1, 2, 3, 5), 1, 8, (1, 1, 1), torch.float32),        # ragged single tile, tiny head
    ((1, 4, 7, 9), 3, 16, (0, 0, 0), torch.float32),       # window of one: out == v
    ((2, 3, 5, 33), 1, 128, (2, 2, 2), torch.float32),
])
def test_attention_vs_oracle(ops, shape, heads, dh, ext, dtype):
    torch.manual_seed(1)
    B, S, H, W = shape
    I = heads * dh
    q, k, v = (torch.randn(B, S, H, W, I).to(dtype).float() for _ in range(3))
    ref, ref_logits = oat.local_attention(k, v, q, ext, heads, return_logits=True)
    out, lse, dbg = ops.local3d_attention_fwd(dev(q).to(dtype), dev(k).to(dtype), dev(v).to(dtype), ext, heads,
                                              need_lse=True, logits_dbg=True)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel(out, ref) < tol
    logits = dbg.cpu().reshape(ref_logits.shape)
    live = ref_logits != -1e9
    assert torch.equal(logits == -1e9, ~live)
    assert rel(logits[live], ref_logits[live]) < 1e-5
    assert torch.allclose(lse.cpu().reshape(-1), torch.logsumexp(ref_logits, -1).reshape(-1), rtol=1e-5, atol=1e-5)
    if ext == (0, 0, 0):
        assert rel(out, v) < tol


def test_attention_strided_qkv(ops):
    """q,k,v as column slices of one fused [N, 3I] buffer (what the fused projection writes)."""
    torch.manual_seed(2)
    B, S, H, W, I = 1, 3, 4, 16, 64
    qkv = torch.randn(B, S, H, W, 3 * I)
    q, k, v = qkv[..., :I], qkv[..., I:2 * I], qkv[..., 2 * I:]
    ref = oat.local_attention(k, v, q, (1, 1, 1), 2)
    d = dev(qkv)
    out, _, _ = ops.local3d_attention_fwd(d[..., :I], d[..., I:2 * I], d[..., 2 * I:], (1, 1, 1), 2)
    assert rel(out, ref) < 1e-5


def test_attention_large_logits_rescale(ops):
    """Force the online-softmax rescale: one key far above the running max late in the walk."""
    torch.manual_seed(3)
    B, S, H, W, I = 1, 7, 16, 16, 32
    q, k, v = (torch.randn(B, S, H, W, I) for _ in range(3))
    k[0, 6, 15, 15] = 40.0 * q[0, 4, 13, 13] / q[0, 4, 13, 13].norm()   # visited last for that query
    k[0, 0, 0, 0] = 30.0 * q[0, 2, 2, 2] / q[0, 2, 2, 2].norm()         # visited first
    ref = oat.local_attention(k, v, q, (3, 3, 3), 1)
    out, _, _ = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), (3, 3, 3), 1)
    assert rel(out, ref) < 1e-5
    assert torch.isfinite(out).all()


# ---------------------------------------------------------------------------------------------- linear

@pytest.mark.parametrize('M,N,K', [(300, 96, 64), (128, 128, 256), (1000, 384, 256), (77, 50, 24), (513, 1024, 256)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_plain_bias_gelu_residual(ops, M, N, K, dtype):
    torch.manual_seed(4)
    a = torch.randn(M, K).to(dtype)
    w = (torch.randn(N, K) / K ** 0.5).to(dtype)
    b = torch.randn(N)
    r = torch.randn(M, N).to(dtype)
    af, wf, rf = a.float(), w.float(), r.float()
    tol = 2e-6 if dtype == torch.float32 else 6e-3
    y = ops.linear_fwd(dev(a), dev(w))
    assert rel(y, af @ wf.t()) < tol
    y = ops.linear_fwd(dev(a), dev(w), bias=dev(b), gelu=True)
    assert rel(y, torch.nn.functional.gelu(af @ wf.t() + b)) < tol
    y = ops.linear_fwd(dev(a), dev(w), bias=dev(b), residual=dev(r))
    assert rel(y, af @ wf.t() + b + rf) < tol
    y = ops.linear_fwd(dev(a), dev(w), bias=dev(b), out_f32=True)
    assert y.dtype == torch.float32 and rel(y, af @ wf.t() + b) < (2e-6 if dtype == torch.float32 else 1e-5)


@pytest.mark.parametrize('M,N,K', [(3072, 1024, 512), (300, 96, 64), (4096, 512, 384), (77, 50, 24), (20000, 256, 256)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_gelu_pair_and_small_row_tiles(ops, M, N, K, dtype):
    """wmz_linear_fwd_gelu_pair (pre-activation + GELU of it from one accumulator; FeedForward's first GEMM in training) and the
    LayerNorm / GELU-prologue GEMMs on 64-row tiles (few rows: config 5) against torch; the pair's second output equals the
    single-output GELU epilogue bit for bit, its first the plain GEMM."""
    torch.manual_seed(41)
    a = (torch.randn(M, K) * 1.5 + 0.3).to(dtype)
    w = (torch.randn(N, K) / K ** 0.5).to(dtype)
    b = torch.randn(N)
    g, be = torch.rand(K) + 0.5, torch.randn(K) * 0.1
    af, wf = a.float(), w.float()
    tol = 3e-6 if dtype == torch.float32 else 8e-3
    for ln in (None, (dev(g), dev(be))):
        xin = af if ln is None else torch.nn.functional.layer_norm(af, (K,), g, be, 1e-5)
        pre = xin @ wf.t() + b
        stats = None if ln is None else ops.layernorm_stats(dev(a), 1e-5)
        z, h = ops.linear_fwd_gelu_pair(dev(a), dev(w), bias=dev(b), ln=ln, ln_stats=stats)
        assert z.dtype == dtype and h.dtype == dtype
        assert rel(z, pre) < tol and rel(h, torch.nn.functional.gelu(pre)) < tol
        assert torch.equal(z, ops.linear_fwd(dev(a), dev(w), bias=dev(b), ln=ln, ln_stats=stats))
        assert torch.equal(h, ops.linear_fwd(dev(a), dev(w), bias=dev(b), ln=ln, ln_stats=stats, gelu=True))
    # the PreNorm GEMM of the training forward in full: + LN(a) as the GEMM consumed it (the weight gradient's plain operand)
    st = ops.layernorm_stats(dev(a), 1e-5)
    c3, h3, an3 = ops.linear_fwd_train(dev(a), dev(w), dev(b), (dev(g), dev(be)), 1e-5, st, want_gelu=True, want_norm=True)
    z3, h3b = ops.linear_fwd_gelu_pair(dev(a), dev(w), bias=dev(b), ln=(dev(g), dev(be)), ln_stats=st)
    assert torch.equal(c3, z3) and torch.equal(h3, h3b) and an3.shape == (M, K) and an3.dtype == dtype
    assert rel(an3, torch.nn.functional.layer_norm(af, (K,), g, be, 1e-5)) < (3e-6 if dtype == torch.float32 else 4e-3)
    c4, h4, an4 = ops.linear_fwd_train(dev(a), dev(w), dev(b), (dev(g), dev(be)), 1e-5, None, want_norm=True)   # statistics computed inside
    assert h4 is None and rel(c4, z3) < tol and rel(an4, an3) < tol
    # GELU in the loader (the backward's fallback when the activation was not kept) on both tile heights
    if N % 8 == 0:
        w2 = (torch.randn(K, N) / N ** 0.5).to(dtype)
        zz = (torch.randn(M, N)).to(dtype)
        y = ops.linear_fwd(dev(zz), dev(w2), gelu_in=True)
        assert rel(y, torch.nn.functional.gelu(zz.float()) @ w2.float().t()) < tol


@pytest.mark.parametrize('M', [3072, 20000, 333])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_wgrad_batch_with_per_problem_layernorm(ops, M, dtype):
    """wmz_linear_wgrad_batch_ln (the op-by-op path's weight gradients of a layer as one launch pair: LayerNorm-prologue and
    plain problems mixed) equals the one-problem launches bit for bit where the slicing of M is the same, and torch in any
    case; few rows take the unsplit form (every tile one workgroup, added straight into dW, accumulate and overwrite)."""
    torch.manual_seed(43)
    shapes = [(512, 512, True), (1536, 512, True), (512, 1024, False), (1024, 512, False), (128, 264, False)]
    probs, refs, singles = [], [], []
    for N, K, ln in shapes:
        dc = (torch.randn(M, N) * 0.3).to(dtype)
        a = (torch.randn(M, K) * 1.3 + 0.2).to(dtype)
        g, b = torch.rand(K) + 0.5, torch.randn(K) * 0.1
        init = torch.randn(N, K)
        binit = torch.randn(N)
        an = torch.nn.functional.layer_norm(a.float(), (K,), g, b, 1e-5) if ln else a.float()
        if ln and dtype == torch.bfloat16:
            an = an.bfloat16().float()                      # the prologue rounds LN(a) to the operand type
        over = N == 1024
        refs.append(((0 if over else init) + dc.float().t() @ an, (0 if over else binit) + dc.float().sum(0)))
        dcd, ad = dev(dc), dev(a)
        stats = ops.layernorm_stats(ad, 1e-5) if ln else (None, None)
        dw, db = dev(init).clone(), dev(binit).clone()
        probs.append(dict(dc=dcd, ldc=N, a=ad, lda=K, dw=dw, dbias=db, M=M, N=N, K=K, g=dev(g) if ln else None,
                          b=dev(b) if ln else None, mean=stats[0], rstd=stats[1], overwrite=over))
        dw1, db1 = dev(init).clone(), dev(binit).clone()
        ops.linear_wgrad(dcd, ad, dw1, db1, ln=(dev(g), dev(b)) if ln else None, ln_stats=stats if ln else None, overwrite=over)
        singles.append((dw1, db1))
    ops.linear_wgrad_batch_ln(probs)
    tol = 3e-5 if dtype == torch.float32 else 2e-3
    for q, (rw, rb), (sw, sb) in zip(probs, refs, singles):
        assert rel(q['dw'], rw) < tol and rel(q['dbias'], rb) < tol
        assert rel(q['dw'], sw.cpu()) < 1e-5 and rel(q['dbias'], sb.cpu()) < 1e-5


@pytest.mark.parametrize('M', [65536, 4099, 300])
def test_wgrad_batch_on_256_wide_tiles(ops, M):
    """wmz_linear_wgrad_batch's bf16 path (wgrad3_kernel: 256 x 256 output tiles, 64 x 128 per wave) against torch and against the
    one-problem launches (version 2's 128 x 128 tiles): the fused path's five shapes, a two-tile problem, a tile used to three
    quarters; ragged M (the last slab of a slice is partly zero rows), accumulate and overwrite, with and without a bias; M too
    small for the big tiles falls back to version 2."""
    torch.manual_seed(47)
    shapes = [(256, 256, True, False), (256, 256, True, True), (256, 128, True, False), (128, 256, False, False), (512, 256, True, False),
              (256, 384, False, True)]
    probs, refs, singles = [], [], []
    for N, K, bias, over in shapes:
        dc = (torch.randn(M, N) * 0.3).bfloat16()
        a = (torch.randn(M, K) * 1.1 + 0.1).bfloat16()
        init, binit = torch.randn(N, K), torch.randn(N)
        refs.append(((0 if over else init) + dc.float().t() @ a.float(), (0 if over else binit) + dc.float().sum(0)))
        dcd, ad = dev(dc), dev(a)
        probs.append((dcd, ad, dev(init).clone(), dev(binit).clone() if bias else None, over))
        dw1, db1 = dev(init).clone(), dev(binit).clone() if bias else None
        ops.linear_wgrad(dcd, ad, dw1, db1, overwrite=over)
        singles.append((dw1, db1))
    ops.linear_wgrad_batch(probs)
    for (dcd, ad, dw, db, over), (rw, rb), (sw, sb) in zip(probs, refs, singles):
        assert rel(dw, rw) < 2e-3 and rel(dw, sw) < 1e-5
        if db is not None:
            assert rel(db, rb) < 2e-3 and rel(db, sb) < 1e-5


@pytest.mark.parametrize('M', [65536, 8231])
def test_wgrad_batch_on_256_wide_tiles_partial_columns(ops, M):
    """wgrad3_kernel's operand ring (round 4: LDS-DMA with the swizzle on the source side) where a 256-wide tile is only partly
    inside the problem: the published width's 128 x 384 / 384 x 128 gradients (3/8 of their tiles: the new eligibility bar), and
    widths that are multiples of 8 only (N = 200: 25 of a tile's 32 chunks; K = 264: the second tile column holds ONE chunk) --
    the out-of-range chunks are fetched from the tile's first chunk and must not reach the result; ragged M on top."""
    torch.manual_seed(53)
    shapes = [(128, 384, True, False), (384, 128, False, True), (200, 264, True, False), (384, 512, True, True), (264, 200, True, False)]
    probs, refs = [], []
    for N, K, bias, over in shapes:
        dc = (torch.randn(M, N) * 0.3).bfloat16()
        a = (torch.randn(M, K) * 1.1 + 0.1).bfloat16()
        init, binit = torch.randn(N, K), torch.randn(N)
        refs.append(((0 if over else init) + dc.float().t() @ a.float(), (0 if over else binit) + dc.float().sum(0)))
        probs.append((dev(dc), dev(a), dev(init).clone(), dev(binit).clone() if bias else None, over))
    ops.linear_wgrad_batch(probs)
    for (dcd, ad, dw, db, over), (rw, rb) in zip(probs, refs):
        assert torch.isfinite(dw).all() and rel(dw, rw) < 2e-3
        if db is not None:
            assert rel(db, rb) < 2e-3


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,S,HW,D,N', [(8, 5, 256, 256, 1024), (3, 4, 20, 32, 50), (2, 1, 77, 64, 130)])
def test_linear_on_last_frame_blocks(ops, dtype, B, S, HW, D, N):
    """logit_proj on x[:, -1] read in place (wmz_linear_fwd_blocked, 64-row tiles for small M): equals the GEMM on a
    gathered copy, and the reference expression on the strided view."""
    torch.manual_seed(6)
    x = torch.randn(B, S, HW, D).to(dtype)
    w = (torch.randn(N, D) / D ** 0.5).to(dtype)
    b = torch.randn(N)
    xd = dev(x)
    last = xd[:, -1]
    assert S == 1 or not last.is_contiguous()
    y = ops.linear_fwd_blocks(last, dev(w), dev(b), out_f32=True)
    ref = x[:, -1].float() @ w.float().t() + b
    assert y.dtype == torch.float32 and y.shape == (B, HW, N)
    assert rel(y, ref) < (2e-6 if dtype == torch.float32 else 1e-5)
    y2 = ops.linear_fwd(last.contiguous(), dev(w), bias=dev(b), out_f32=True)
    assert torch.equal(y, y2)
    yb = ops.linear_fwd_blocks(last, dev(w), dev(b))
    assert yb.dtype == dtype and rel(yb, ref) < (2e-6 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_layernorm_prologue(ops, dtype):
    torch.manual_seed(5)
    M, N, K = 333, 256, 256
    a = (torch.randn(M, K) * 2 + 0.7).to(dtype)
    w = (torch.randn(N, K) / K ** 0.5).to(dtype)
    gam, bet, b = torch.rand(K) + 0.5, torch.randn(K) * 0.1, torch.randn(N)
    ln = torch.nn.functional.layer_norm(a.float(), (K,), gam, bet, 1e-5)
    if dtype == torch.bfloat16:
        ln = ln.to(dtype).float()       # the kernel rounds LN(x) to the MFMA operand type
    ref = ln @ w.float().t() + b
    y = ops.linear_fwd(dev(a), dev(w), bias=dev(b), ln=(dev(gam), dev(bet)))
    assert rel(y, ref) < (3e-6 if dtype == torch.float32 else 6e-3)


def test_embed_pos3d(ops):
    g = load_golden('transformer_tiny')
    sd = sub(g, 'sd/')
    ref = oden.embed_tokens(sd, g['z'])
    x = ops.embed_pos3d_fwd(dev(g['z']), dev(sd['transformer.embedding.weight']),
                            dev(sd['transformer.pos_emb_s.weight']), dev(sd['transformer.pos_emb_h.weight']),
                            dev(sd['transformer.pos_emb_w.weight']), torch.float32)
    assert torch.equal(x.cpu(), ref)                      # same fp32 additions in the same order
    xb = ops.embed_pos3d_fwd(dev(g['z']), dev(sd['transformer.embedding.weight']),
                             dev(sd['transformer.pos_emb_s.weight']), dev(sd['transformer.pos_emb_h.weight']),
                             dev(sd['transformer.pos_emb_w.weight']), torch.bfloat16)
    assert torch.equal(xb.cpu(), ref.to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------- VQ

@pytest.mark.parametrize('name', ['vq_encode_512', 'vq_encode_1024', 'vq_encode_8192', 'vq_encode_odd'])
def test_vq_argmin_bit_exact_vs_golden(ops, name):
    g = load_golden(name)
    cb = g['embedding'][0]
    idx, dmin = ops.vq_argmin(dev(g['x']), dev(cb), need_dist=True)
    assert idx.dtype == torch.int64
    assert torch.equal(idx.cpu(), g['idx'][:, 0])                        # bit-identical indices
    if 'dist_min' in g:
        assert torch.equal(dmin.cpu(), g['dist_min'])                    # and bit-identical distances
    else:
        assert torch.equal(dmin.cpu(), g['dist'].min(dim=-1).values)
    dec = ops.vq_gather(idx, dev(cb))
    assert torch.equal(dec.cpu(), ovq.decode(g['idx'], g['embedding'])[:, 0])


@pytest.mark.parametrize('N,C,E', [(65536, 1024, 64), (4097, 512, 64), (1000, 8192, 64), (999, 100, 32),
                                   (500, 77, 24), (64, 33, 40), (129, 16, 8)])
def test_vq_argmin_vs_oracle(ops, N, C, E):
    """Full-size encode stage of configs 3/4 (N = 65 536, C = 1024) and ragged shapes: indices and min
    distances bit-identical to the oracle (which evaluates the reference's own tensor expression)."""
    torch.manual_seed(6)
    x, cb = torch.randn(N, E), torch.randn(C, E)
    ref = ovq.distances(x, cb[None])[:, 0]
    idx, dmin = ops.vq_argmin(dev(x), dev(cb), need_dist=True)
    assert torch.equal(idx.cpu(), ref.argmin(-1))
    assert torch.equal(dmin.cpu(), ref.min(-1).values)


@pytest.mark.parametrize('case', ['gauss', 'duplicates', 'clustered', 'tiny', 'huge', 'rows_on_codes', 'nan_row', 'mixed_scale'])
def test_vq_screened_argmin_equals_the_exact_scan(ops, case):
    """embedding_dim 64 / multiple-of-64 codes take the screened search (matrix-core screening with a proven error bound +
    exact re-check of undecided rows, csrc/vq_screen.hip).  It must return EXACTLY what the full scan in the pinned fp32 order
    returns -- indices and minimum distances, ties to the lowest index -- on data built to stress the bound: duplicated codes,
    codebooks clustered far tighter than the bound, rows lying on codes, tiny / huge magnitudes, a NaN row."""
    g = torch.Generator().manual_seed(11)
    N, C, E = 5000, 256, 64
    x, cb = torch.randn(N, E, generator=g), torch.randn(C, E, generator=g)
    if case == 'duplicates':
        cb[100] = cb[7]; cb[200] = cb[7]; cb[201] = cb[13]
        x[:64] = cb[7] + 1e-3 * torch.randn(64, E, generator=g)
    elif case == 'clustered':                    # 8 clusters of 32 codes 1e-4 apart: every row's best codes are near ties
        centers = torch.randn(8, E, generator=g)
        cb = centers.repeat_interleave(32, 0) + 1e-4 * torch.randn(C, E, generator=g)
    elif case == 'tiny':
        x, cb = x * 1e-3, cb * 1e-3
    elif case == 'huge':
        x, cb = x * 300.0, cb * 300.0
    elif case == 'rows_on_codes':
        x[:C] = cb
    elif case == 'nan_row':
        x[5, 3] = float('nan')
    elif case == 'mixed_scale':                  # one huge code sets emax: the bound widens for every row
        cb[9] = cb[9] * 1e3
    xd, cbd = dev(x), dev(cb)
    idx_s, d_s = ops.vq_argmin(xd, cbd, need_dist=True)
    idx_e, d_e = ops.vq_argmin(xd, cbd, need_dist=True, exact_scan=True)
    assert torch.equal(idx_s, idx_e)
    assert torch.equal(torch.nan_to_num(d_s, nan=-1.0), torch.nan_to_num(d_e, nan=-1.0))
    if case not in ('nan_row',):
        ref = ovq.distances(x, cb[None])[:, 0]
        assert torch.equal(idx_s.cpu(), ref.argmin(-1)) and torch.equal(d_s.cpu(), ref.min(-1).values)


def test_vq_screened_argmin_rechecks_few_rows_on_gaussian_data(ops):
    """The share of rows the screening cannot decide (they cost a whole-codebook scan each) on the SURVEY 8(d) micro-bench data."""
    from world_modelz_amd import ops as O
    g = torch.Generator().manual_seed(0)
    x, cb = dev(torch.randn(65536, 64, generator=g)), dev(torch.randn(1024, 64, generator=g))
    O.vq_argmin(x, cb)
    torch.cuda.synchronize()
    nflag = int(O._vq_screen_ws[x.device][4:8].view(torch.int32).item())       # workspace header: [emax^2 bits, flag count]
    print(f'[vq screened] {nflag} of 65536 rows re-scanned ({100.0 * nflag / 65536:.2f} %)')
    assert nflag < 65536 // 50


def test_vq_ties_and_duplicates(ops):
    torch.manual_seed(7)
    cb = torch.randn(40, 16)
    cb[33] = cb[2]
    cb[17] = cb[2]
    x = cb[[2, 33, 17, 5]].clone()
    idx = ops.vq_argmin(dev(x), dev(cb))
    assert idx.cpu().tolist() == [2, 2, 2, 5]
    # all-equal codebook: index 0
    idx = ops.vq_argmin(dev(torch.randn(10, 16)), dev(torch.ones(7, 16)))
    assert idx.cpu().tolist() == [0] * 10


def test_vq_forward_sequence_vs_golden(ops):
    """EMA statistics + update kernels reproduce the reference's buffer trajectory (quirk Q4)."""
    g = load_golden('vq_forward_train')
    emb = dev(g['embedding0'][0].clone())
    cs = dev(g['cluster_size0'][0].clone())
    act = torch.zeros(32, device='cuda')
    err = torch.zeros(32, device='cuda')
    for tag in ['t0', 't1', 't2']:
        x = dev(g[f'{tag}/x'])
        idx = ops.vq_argmin(x, emb)
        assert torch.equal(idx.cpu(), g[f'{tag}/encodings_argmax'][:, 0])
        counts = torch.zeros(32, device='cuda')
        dw = torch.zeros(32, 8, device='cuda')
        ops.vq_ema_stats(x, idx, emb, counts, dw, err)
        ops.vq_ema_update(emb, cs, act, counts, dw, 0.99, 1e-5)
        assert torch.allclose(emb.cpu(), g[f'{tag}/embedding'][0], rtol=1e-5, atol=1e-6)
        assert torch.allclose(cs.cpu(), g[f'{tag}/cluster_size'][0], rtol=1e-6)
        assert torch.equal(act.cpu(), g[f'{tag}/activation_count'][0])
        assert torch.allclose(err.cpu(), g[f'{tag}/accumulated_error'][0], rtol=1e-5)


def test_library_refuses_cpu_tensors(ops):
    from world_modelz_amd._lib import WmzError
    with pytest.raises(WmzError):
        ops.vq_argmin(torch.randn(4, 8), torch.randn(3, 8))


@pytest.mark.parametrize('shape,heads,dh,ext', [
    ((2, 5, 16, 16), 1, 128, (3, 3, 3)),       # BASELINE plane shape
    ((1, 9, 16, 16), 1, 128, (3, 1, 1)),       # published run-03 window (7,3,3)
    ((1, 4, 16, 16), 2, 64, (1, 2, 3)),
    ((1, 3, 40, 16), 1, 32, (2, 2, 2)),        # H > 16: several workgroups per plane
    ((1, 2, 5, 16), 4, 32, (0, 1, 0)),         # H < 16: idle waves
    ((1, 6, 16, 16), 1, 128, (5, 0, 7)),       # wide column window, single row
    ((2, 2, 1, 16), 4, 128, (0, 1, 16)),       # ONE row per plane (config 5's dense attention over 16 tokens): no slab of odd rows
    ((1, 2, 17, 16), 1, 64, (1, 2, 2)),        # H = 1 (mod 16): the last chunk has no odd row either
])
def test_attention_row16_fast_path(ops, shape, heads, dh, ext):
    """bf16, W == 16 takes attn_fwd_row16.hip -- the kernel bench.py times.  Probed directly (its own logits dump, not the
    general kernel's): scaled logits <= 1e-3 rel against the fp32 oracle on the same bf16 inputs (north_star bar; measured
    ~1e-6, bf16 products are exact in the fp32 accumulator), masked slots the literal -1e9, lse <= 1e-4, out <= 3e-3 rel
    (P and the output are rounded to bf16: 2^-9 per element), and against the general kernel."""
    torch.manual_seed(21)
    B, S, H, W = shape
    I = heads * dh
    q, k, v = (torch.randn(B, S, H, W, I).bfloat16() for _ in range(3))
    ref, ref_logits = oat.local_attention(k.float(), v.float(), q.float(), ext, heads, return_logits=True)
    fast, lse_f, dbg = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), ext, heads, need_lse=True, logits_dbg=True)
    plain, _, _ = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), ext, heads)        # the un-probed instantiation
    gen, lse_g, _ = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), ext, heads, need_lse=True, general=True)
    assert torch.equal(fast, plain)
    logits = dbg.cpu().reshape(ref_logits.shape)
    live = ref_logits != -1e9
    assert torch.equal(logits == -1e9, ~live)
    err = float((logits[live] - ref_logits[live]).abs().max() / ref_logits[live].abs().max())
    assert err < 1e-3 and rel(logits[live], ref_logits[live]) < 1e-3, err
    e_out, e_gen = rel(fast, ref), rel(fast, gen)
    print(f'[row16 {shape} {ext}] logits max err {err:.1e}, out vs oracle {e_out:.2e}, vs general kernel {e_gen:.2e}')
    assert e_out < 3e-3
    assert e_gen < 3e-3
    lse_ref = torch.logsumexp(ref_logits, -1).reshape(-1, heads)
    assert torch.allclose(lse_f.cpu(), lse_ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(lse_f.cpu(), lse_g.cpu(), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('shape,heads,dh,ext', [
    ((1, 8, 8, 8), 1, 128, (3, 3, 3)),         # BASELINE configs[1]: 8x8x8 latent grid
    ((2, 6, 8, 8), 1, 128, (3, 1, 1)),         # the reference's own geometry (main.py:394) with the published window (7,3,3)
    ((1, 3, 8, 8), 2, 64, (1, 2, 2)),          # even row extent: both rim rows admit two of the three sub-row offsets
    ((1, 2, 8, 8), 1, 32, (1, 0, 3)),          # eH = 0: the only tile row is a rim row
    ((1, 3, 16, 8), 1, 128, (1, 3, 1)),        # 8 tile rows: two 4-wave workgroups per plane
    ((1, 2, 24, 8), 1, 64, (0, 5, 2)),         # 12 tile rows: the 16-wave shape with 8-wide planes
    ((1, 2, 6, 8), 1, 128, (1, 1, 1)),         # 3 tile rows: a ragged chunk
    ((1, 2, 8, 8), 1, 128, (1, 9, 9)),         # window larger than the plane
    ((2, 3, 2, 8), 1, 128, (1, 1, 3)),         # two plane rows = ONE tile row
])
def test_attention_8_wide_planes_fast_path(ops, shape, heads, dh, ext):
    """bf16, W == 8 with an even number of rows takes attn_fwd_row16.hip too (round 4): the plane is read as H / 2 tile rows of 16,
    the window test runs in tile coordinates with per-lane sub-row masks in the rim rows.  Same bar as the 16-wide test: the
    kernel's own logits dump against the fp32 oracle (<= 1e-3, masked slots the literal -1e9), lse, out, and the general kernel."""
    torch.manual_seed(23)
    B, S, H, W = shape
    I = heads * dh
    q, k, v = (torch.randn(B, S, H, W, I).bfloat16() for _ in range(3))
    ref, ref_logits = oat.local_attention(k.float(), v.float(), q.float(), ext, heads, return_logits=True)
    fast, lse_f, dbg = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), ext, heads, need_lse=True, logits_dbg=True)
    plain, _, _ = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), ext, heads)
    gen, lse_g, _ = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), ext, heads, need_lse=True, general=True)
    assert torch.equal(fast, plain)
    logits = dbg.cpu().reshape(ref_logits.shape)
    live = ref_logits != -1e9
    assert torch.equal(logits == -1e9, ~live)
    err = float((logits[live] - ref_logits[live]).abs().max() / ref_logits[live].abs().max())
    assert err < 1e-3 and rel(logits[live], ref_logits[live]) < 1e-3, err
    e_out, e_gen = rel(fast, ref), rel(fast, gen)
    print(f'[8-wide {shape} {ext}] logits max err {err:.1e}, out vs oracle {e_out:.2e}, vs general kernel {e_gen:.2e}')
    assert e_out < 3e-3 and e_gen < 3e-3
    lse_ref = torch.logsumexp(ref_logits, -1).reshape(-1, heads)
    assert torch.allclose(lse_f.cpu(), lse_ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(lse_f.cpu(), lse_g.cpu(), rtol=1e-5, atol=2e-5)


def test_attention_row16_deferred_max_branch(ops):
    """Force both sides of the deferred-rescale decision: a key far above the running max late in the walk (rescale
    must fire) and logits that creep up by < 2^8 per step (rescale deferred; P may exceed 1)."""
    torch.manual_seed(22)
    B, S, H, W, I = 1, 7, 16, 16, 128
    q, k, v = (torch.randn(B, S, H, W, I) * 0.5 for _ in range(3))
    k[0, 6, 15, 15] = 60.0 * q[0, 4, 13, 13] / q[0, 4, 13, 13].norm()     # visited last for that query: big jump
    for i in range(7):                                                     # slowly growing logits for query (3,8,8)
        k[0, i, 8, 8] = (2.0 + 2.5 * i) * q[0, 3, 8, 8] / q[0, 3, 8, 8].norm() * (128 ** 0.5) / q[0, 3, 8, 8].norm()
    q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    ref = oat.local_attention(k.float(), v.float(), q.float(), (3, 3, 3), 1)
    out, _, _ = ops.local3d_attention_fwd(dev(q), dev(k), dev(v), (3, 3, 3), 1)
    assert torch.isfinite(out).all()
    assert rel(out, ref) < 3e-3
    for pos in [(0, 4, 13, 13), (0, 3, 8, 8)]:
        assert rel(out[pos], ref[pos]) < 1e-2
    # very negative logits everywhere (the first step must SET the running max, not assume 0): q . k ~ -3000 / sqrt(128)
    qn = torch.full((1, 3, 16, 16, 128), 5.0).bfloat16()
    kn = (-torch.full((1, 3, 16, 16, 128), 5.0) + 0.05 * torch.randn(1, 3, 16, 16, 128)).bfloat16()
    vn = torch.randn(1, 3, 16, 16, 128).bfloat16()
    refn = oat.local_attention(kn.float(), vn.float(), qn.float(), (1, 1, 1), 1)
    outn, lsen, _ = ops.local3d_attention_fwd(dev(qn), dev(kn), dev(vn), (1, 1, 1), 1, need_lse=True)
    assert torch.isfinite(outn).all() and torch.isfinite(lsen).all() and rel(outn, refn) < 5e-3


def test_attention_full_size_properties(ops):
    """BASELINE configs[3] size (8 x 32 x 16 x 16 tokens, dim_head 128, window 7x7x7), where the CPU oracle takes too long
    for a unit test: size-independent properties of softmax attention instead.
      * V constant per feature  -> every output row equals that constant (the weights of a row sum to 1);
      * linearity in V: attn(q, k, a*V1 + b*V2) = a*attn(q, k, V1) + b*attn(q, k, V2) (same weights);
      * one clip of the batch equals the same clip run alone (clips are independent: the data-parallel sharding);
      * translation along the time axis away from the clip ends: shifting q, k, v by 2 eS + 1 = 7 planes shifts the output
        bit for bit (the kernel visits a window's key planes in an order rotated by the plane index mod 7, so that the
        workgroups sharing a plane stage it together: a shift by 7 keeps every summation order), a shift by one plane shifts
        it up to fp32 summation order."""
    torch.manual_seed(31)
    B, S, H, W, I = 8, 32, 16, 16, 128
    ext = (3, 3, 3)
    q, k = (torch.randn(B, S, H, W, I, device='cuda').bfloat16() for _ in range(2))
    v1, v2 = (torch.randn(B, S, H, W, I, device='cuda').bfloat16() for _ in range(2))

    def attn(qq, kk, vv):
        return ops.local3d_attention_fwd(qq, kk, vv, ext, 1)[0].float()
    const = torch.randn(I, device='cuda').bfloat16()
    out_c = attn(q, k, const.expand(B, S, H, W, I).contiguous())
    assert (out_c - const.float()).abs().max() < 2e-2 * const.float().abs().max()
    o1, o2 = attn(q, k, v1), attn(q, k, v2)
    o12 = attn(q, k, (0.5 * v1.float() - 0.25 * v2.float()).bfloat16())
    lin = 0.5 * o1 - 0.25 * o2
    assert float((o12 - lin).norm() / lin.norm()) < 1.5e-2            # bf16 rounding of the mixed V and of the outputs
    alone = attn(q[3:4].contiguous(), k[3:4].contiguous(), v1[3:4].contiguous())
    assert torch.equal(alone[0], o1[3])
    shifted = attn(q[:, 7:].contiguous(), k[:, 7:].contiguous(), v1[:, 7:].contiguous())
    # planes whose window does not touch either end of either clip see exactly the same keys in the same order
    assert torch.equal(shifted[:, 3:S - 7 - 3], o1[:, 10:S - 3])
    shifted1 = attn(q[:, 1:].contiguous(), k[:, 1:].contiguous(), v1[:, 1:].contiguous())
    e1 = float((shifted1[:, 3:S - 5] - o1[:, 4:S - 4]).norm() / o1[:, 4:S - 4].norm())
    print(f'[attention full size] shift by one plane: rel {e1:.2e} (P is rounded to bf16 against another running reference)')
    assert e1 < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize('N,C,E,dominant', [(65536, 1024, 64, False), (5000, 37, 16, True), (777, 512, 96, True), (300, 8, 200, False)])
def test_vq_ema_statistics_sorted_gather_vs_scatter_and_torch(N, C, E, dominant):
    """wmz_vq_ema_stats_sorted (counting sort by code + gather) against index_add in float64 and against the scatter kernel:
    counts exact, dw / sqerr to fp32 summation order; accumulates into non-zero buffers; two calls in a row (the sort's
    counters must be back at zero); a dominant code; row counts that are not a multiple of the 64-entry wave tile; E that is
    not a multiple of 64; out-of-range indices clamped."""
    from world_modelz_amd import ops, _lib as L
    torch.manual_seed(9)
    x = torch.randn(N, E, device='cuda')
    cb = torch.randn(C, E, device='cuda')
    idx = torch.randint(0, C, (N,), device='cuda')
    if dominant:
        idx[torch.rand(N, device='cuda') < 0.6] = C // 3
        idx[0], idx[1] = -4, C + 11
    c0, d0, s0 = torch.rand(C, device='cuda'), torch.randn(C, E, device='cuda'), torch.rand(C, device='cuda')
    counts, dw, sq = c0.clone(), d0.clone(), s0.clone()
    ops.vq_ema_stats(x, idx, cb, counts, dw, sq)
    ops.vq_ema_stats(x, idx, cb, counts, dw, sq)
    ic = idx.clamp(0, C - 1)
    rc = c0.double() + 2 * torch.bincount(ic, minlength=C).double()
    rd = d0.double().index_add_(0, ic, 2 * x.double())
    rs = s0.double().index_add_(0, ic, 2 * ((cb[ic].double() - x.double()) ** 2).sum(1))
    assert torch.equal(counts.double(), rc) or float((counts.double() - rc).abs().max()) < 1e-3     # integers, + a random start
    assert float((dw.double() - rd).abs().max() / rd.abs().max()) < 2e-6
    assert float((sq.double() - rs).abs().max() / rs.abs().max()) < 2e-6
    # the scatter kernel (> 12 288 codes) computes the same
    counts2, dw2, sq2 = c0.clone(), d0.clone(), s0.clone()
    for _ in range(2):
        L.call('wmz_vq_ema_stats', L.ptr(x), E, L.ptr(idx), L.ptr(cb), L.ptr(counts2), L.ptr(dw2), L.ptr(sq2), N, C, E, L.stream())
    torch.cuda.synchronize()
    # (its thousands of sequential fp32 atomic adds into one dominant code's sums are the LESS accurate of the two)
    assert float((dw2 - dw).abs().max() / rd.abs().max()) < 1e-5 and float((sq2 - sq).abs().max() / rs.abs().max()) < 1e-4


@pytest.mark.parametrize('n_lat', [2, 3])
def test_vector_quantizer_with_several_latents_vs_oracle(n_lat):
    """VectorQuantizerEMA(num_latents > 1) (vq.py:7, :27: every latent has its own codebook; the reference's callers in scope use 1 --
    train_vqae.py:31 -- but the constructor accepts it): encode / decode / three training forwards + one evaluation forward against
    the oracle's restatement of vq.py:25-94 -- indices bit-identical, every buffer, the quantised tensor, loss, perplexity, the
    dense one-hot on request, and the straight-through gradient."""
    from oracle import vq as ovq
    from world_modelz_amd.vq import VectorQuantizerEMA
    torch.manual_seed(17)
    E, C = 8, 32
    m = VectorQuantizerEMA(E, C, num_latents=n_lat).cuda()
    st = {'embedding': m.embedding.detach().cpu().clone(), 'cluster_size': m.cluster_size.detach().cpu().clone(),
          'activation_count': torch.zeros(n_lat, C), 'accumulated_error': torch.zeros(n_lat, C)}
    x0 = torch.randn(6, 10, n_lat * E)
    assert torch.equal(m.encode(x0.cuda()).cpu(), ovq.encode(x0, st['embedding']))
    idx = ovq.encode(x0, st['embedding']).view(6, 10, n_lat)
    assert torch.equal(m.decode(idx.cuda()).cpu(), ovq.decode(idx, st['embedding']))
    for step in range(4):
        training = step < 3
        m.train(training)
        x = torch.randn(6, 10, n_lat * E)
        xg = x.cuda().requires_grad_(True)
        q, enc, loss, ppl = m(xg)
        q_ref, enc_ref, loss_ref, ppl_ref = ovq.forward(x, st, training)
        assert q.shape == x.shape and torch.allclose(q.detach().cpu(), q_ref, rtol=1e-6, atol=1e-6)
        assert torch.allclose(loss.detach().cpu(), loss_ref, rtol=1e-5) and torch.allclose(ppl.cpu(), ppl_ref, rtol=1e-5)
        assert enc.shape == enc_ref.shape and torch.equal(enc.indices.cpu(), enc_ref.argmax(-1))
        if step == 0:
            assert torch.equal(enc.materialize().cpu(), enc_ref)
        for name in ('embedding', 'cluster_size', 'activation_count', 'accumulated_error'):
            assert torch.allclose(getattr(m, name).cpu(), st[name], rtol=1e-5, atol=1e-6), (step, name)
        (q * 2.0).sum().backward()
        assert torch.equal(xg.grad.cpu(), torch.full_like(x, 2.0))             # straight-through: d quantized / d input = 1
    m.activation_count[1, :5] = 0
    st['activation_count'][1, :5] = 0
    before = m.embedding.cpu().clone()
    dead = (m.activation_count == 0).cpu()
    n = m.reuse_inactive()
    assert n == ovq.reuse_inactive(st) == int(dead.sum()) >= 5             # (which of several equally active codes topk names is
    moved = (m.embedding.cpu() - before).abs().amax(-1) > 0                 #  the device's choice: only the count and the rows compared)
    assert torch.equal(moved, dead)
