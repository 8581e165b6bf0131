"""GPU: the fused per-token backward kernels (csrc/layer_fused_bwd.hip) against a plain fp32 torch restatement of the same
math (reference local_3d_attention.py:11-31 PreNorm / FeedForward, :46-53 to_q / to_k / to_v / to_out), through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

D, I, M = 256, 128, 256


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


@pytest.fixture(scope='module')
def layer():
    from world_modelz_amd import config
    from world_modelz_amd.local_3d_attention import Local3dAttentionTransformer
    torch.manual_seed(5)
    tr = Local3dAttentionTransformer(data_shape=(3, 16, 16), dim=D, num_classes=64, extents=(1, 1, 1), depth=1, mlp_dim=M,
                                     dim_head=I, heads=1).cuda()
    with torch.no_grad():                                  # non-trivial LayerNorm affines and biases
        for p in tr.parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    return tr.layers[0]


def _tile_z(z):
    """[ntok, M] -> the tiled layout of layer_fused.hip (per 32-token tile [M/32 chunks][2][64 lanes = (h, t)][8])."""
    n = z.shape[0]
    v = z.reshape(n // 32, 32, M // 32, 2, 2, 8)           # tile, t, c, h, j, e
    return v.permute(0, 2, 4, 3, 1, 5).contiguous().reshape(n, M)


def _ln_bwd(dxhat, x, mean, rstd):
    xh = (x - mean[:, None]) * rstd[:, None]
    return rstd[:, None] * (dxhat - dxhat.mean(1, keepdim=True) - xh * (dxhat * xh).mean(1, keepdim=True)), xh


@pytest.mark.parametrize('n', [2 * 16 * 16 * 3, 768 + 96])     # 1536 = 6 full workgroups; 864 = 3 + one with 3 of 8 waves
def test_ff_fused_bwd_vs_torch(layer, n):
    from world_modelz_amd import _lib as L, fused
    attn, ff = layer
    torch.manual_seed(1)
    dy = torch.randn(n, D, device='cuda').bfloat16()
    x1 = (torch.randn(n, D, device='cuda') * 1.5 + 0.2).bfloat16()
    xf = x1.float()
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    st = torch.stack([mean, rstd]).contiguous()
    g2, be2 = ff.norm.weight.detach(), ff.norm.bias.detach()
    w1, b1, w2 = ff.fn.net[0].weight.detach(), ff.fn.net[0].bias.detach(), ff.fn.net[3].weight.detach()
    wout = attn.fn.to_out[0].weight.detach()
    xh = (xf - mean[:, None]) * rstd[:, None]
    z = ((xh * g2 + be2) @ w1.t() + b1).bfloat16()
    zt = _tile_z(z)
    _, wpack_ff = fused._layer_pack_bwd(attn, ff)
    outs = [torch.empty(n, w, dtype=torch.bfloat16, device='cuda') for w in (M, M, D, D, I)]
    g, dz, xhat1, dx1, do = outs
    L.call('wmz_ff_fused_bwd', L.ptr(dy), L.ptr(zt), L.ptr(x1), L.ptr(st), L.ptr(g), L.ptr(dz), L.ptr(xhat1), L.ptr(dx1),
           L.ptr(do), L.ptr(wpack_ff), n, D, I, M, 0, 0, None, L.stream())
    torch.cuda.synchronize()
    # last-plane mode (the denoiser's loss reads x[:, -1] only): dy restricted to the clips' last planes, every other row
    # read from a zero row -- must equal the dense call on the zero-padded gradient to the bit
    S_, HW_ = 3, n // 6
    dy_pad = dy.clone().view(-1, S_, HW_, D)
    dy_pad[:, :-1] = 0
    outs_a = [torch.empty_like(t) for t in outs]
    outs_b = [torch.empty_like(t) for t in outs]
    L.call('wmz_ff_fused_bwd', L.ptr(dy_pad), L.ptr(zt), L.ptr(x1), L.ptr(st), *[L.ptr(t) for t in outs_a], L.ptr(wpack_ff),
           n, D, I, M, 0, 0, None, L.stream())
    dy_last = dy_pad[:, -1].contiguous()
    zero_row = torch.zeros(D, dtype=torch.bfloat16, device='cuda')
    L.call('wmz_ff_fused_bwd', L.ptr(dy_last), L.ptr(zt), L.ptr(x1), L.ptr(st), *[L.ptr(t) for t in outs_b], L.ptr(wpack_ff),
           n, D, I, M, S_, HW_, L.ptr(zero_row), L.stream())
    torch.cuda.synchronize()
    for a_, b_ in zip(outs_a, outs_b):
        assert torch.equal(a_, b_)
    zf = z.float().requires_grad_(True)
    gr = torch.nn.functional.gelu(zf)
    (dgelu,) = torch.autograd.grad(gr.sum(), zf)
    w2b, w1b, woutb = w2.bfloat16().float(), (w1 * g2).bfloat16().float(), wout.bfloat16().float()
    dz_r = (dy.float() @ w2b) * dgelu
    dxhat = dz_r.bfloat16().float() @ w1b
    lnb, xh_r = _ln_bwd(dxhat, xf, mean, rstd)
    dx1_r = dy.float() + lnb
    do_r = dx1_r.bfloat16().float() @ woutb
    errs = dict(g=rel(g, gr), dz=rel(dz, dz_r), xhat=rel(xhat1, xh_r), dx1=rel(dx1, dx1_r), do=rel(do, do_r))
    print('[ff_fused_bwd]', {k: f'{v:.2e}' for k, v in errs.items()})
    assert all(v < 6e-3 for v in errs.values()), errs
    # the training path: the forward stored the NORMALISED rows (WMZ_FUSED_X1_NORMALISED) -- they come in as `x1`, xhat_out is
    # NULL, nothing is recomputed or written
    outs_n = [torch.empty_like(t_) for t_ in outs]
    ptrs = [L.ptr(t_) for t_ in outs_n]
    ptrs[2] = None
    L.call('wmz_ff_fused_bwd', L.ptr(dy), L.ptr(zt), L.ptr(xh_r.bfloat16().contiguous()), L.ptr(st), *ptrs, L.ptr(wpack_ff),
           n, D, I, M, 0, 0, None, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(outs_n[0], g) and torch.equal(outs_n[1], dz)
    e_n = dict(dx1=rel(outs_n[3], dx1_r), do=rel(outs_n[4], do_r))
    print('[ff_fused_bwd, normalised input]', {k: f'{v:.2e}' for k, v in e_n.items()})
    assert all(v < 6e-3 for v in e_n.values()), e_n


@pytest.mark.parametrize('with_res,n', [(True, 1536), (False, 1536), (True, 864)])
def test_qkv_fused_bwd_vs_torch(layer, with_res, n):
    from world_modelz_amd import _lib as L, fused
    attn, ff = layer
    torch.manual_seed(2)
    dq = torch.randn(n, I, device='cuda').bfloat16()
    dkv = torch.randn(n, 2 * I, device='cuda').bfloat16()
    x = (torch.randn(n, D, device='cuda') * 0.7 - 0.1).bfloat16()
    res = torch.randn(n, D, device='cuda').bfloat16() if with_res else None
    xf = x.float()
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    st = torch.stack([mean, rstd]).contiguous()
    g1 = attn.norm.weight.detach()
    wq, wk, wv = (attn.fn.to_q.weight.detach(), attn.fn.to_k.weight.detach(), attn.fn.to_v.weight.detach())
    wpack_qkv, _ = fused._layer_pack_bwd(attn, ff)
    dx = torch.empty(n, D, dtype=torch.bfloat16, device='cuda')
    xhat = torch.empty(n, D, dtype=torch.bfloat16, device='cuda')
    L.call('wmz_qkv_fused_bwd', L.ptr(dq), I, L.ptr(dkv), 2 * I, L.ptr(x), L.ptr(st), L.ptr(res), L.ptr(dx), L.ptr(xhat),
           L.ptr(wpack_qkv), n, D, I, L.stream())
    torch.cuda.synchronize()
    dxhat = dkv[:, :I].float() @ (wk * g1).bfloat16().float() + dkv[:, I:].float() @ (wv * g1).bfloat16().float()
    lnb, xh_r = _ln_bwd(dxhat, xf, mean, rstd)
    dx_r = lnb + dq.float() @ wq.bfloat16().float()
    if with_res:
        dx_r = dx_r + res.float()
    e1, e2 = rel(dx, dx_r), rel(xhat, xh_r)
    print(f'[qkv_fused_bwd res={with_res}] dx {e1:.2e} xhat {e2:.2e}')
    assert e1 < 6e-3 and e2 < 6e-3
    # the training path on a tiled stream: `x` holds the NORMALISED rows (WMZ_FUSED_XRM_NORMALISED), xhat_out is NULL
    dx_n = torch.empty_like(dx)
    L.call('wmz_qkv_fused_bwd', L.ptr(dq), I, L.ptr(dkv), 2 * I, L.ptr(xh_r.bfloat16().contiguous()), L.ptr(st), L.ptr(res),
           L.ptr(dx_n), None, L.ptr(wpack_qkv), n, D, I, L.stream())
    torch.cuda.synchronize()
    e3 = rel(dx_n, dx_r)
    print(f'[qkv_fused_bwd res={with_res}, normalised input] dx {e3:.2e}')
    assert e3 < 6e-3


def test_ln_affine_grads_vs_torch():
    from world_modelz_amd import _lib as L
    torch.manual_seed(3)
    N, K = 256, 256
    G, W = torch.randn(N, K, device='cuda'), torch.randn(N, K, device='cuda')
    s, gamma, beta = torch.randn(N, device='cuda'), torch.randn(K, device='cuda'), torch.randn(K, device='cuda')
    dW, dbias = torch.randn(N, K, device='cuda'), torch.randn(N - 128, device='cuda')
    dg, db = torch.randn(K, device='cuda'), torch.randn(K, device='cuda')
    ref = (dW + G * gamma + s[:, None] * beta, dbias + s[128:], dg + (W * G).sum(0), db + W.t() @ s)
    L.call('wmz_ln_affine_grads', L.ptr(G), L.ptr(s), L.ptr(W), L.ptr(gamma), L.ptr(beta), L.ptr(dW), L.ptr(dbias), L.ptr(dg),
           L.ptr(db), N, K, 128, L.stream())
    for a, b in zip((dW, dbias, dg, db), ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
    # the batched call: two problems of different shapes in one launch (the grid covers the larger; dbias may be NULL)
    from world_modelz_amd import fused
    N2, K2 = 72, 128
    G2, W2 = torch.randn(N2, K2, device='cuda'), torch.randn(N2, K2, device='cuda')
    s2, ga2, be2 = torch.randn(N2, device='cuda'), torch.randn(K2, device='cuda'), torch.randn(K2, device='cuda')
    dW1, dbias1, dg1, db1 = (torch.zeros_like(t) for t in (dW, dbias, dg, db))
    dW2, dg2, db2 = torch.zeros(N2, K2, device='cuda'), torch.zeros(K2, device='cuda'), torch.zeros(K2, device='cuda')
    fused._ln_affine_grads_batch([(G, s, W, gamma, beta, dW1, dbias1, dg1, db1, N, K, 128),
                                  (G2, s2, W2, ga2, be2, dW2, None, dg2, db2, N2, K2, 0)])
    ref1 = (G * gamma + s[:, None] * beta, s[128:], (W * G).sum(0), W.t() @ s)
    ref2 = (G2 * ga2 + s2[:, None] * be2, (W2 * G2).sum(0), W2.t() @ s2)
    for a, b in zip((dW1, dbias1, dg1, db1, dW2, dg2, db2), ref1 + ref2):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)


def test_pack_set_equals_the_per_stream_packers():
    """fused.PackSet (one table-driven call for every stream of the model) fills exactly what wmz_layer_fused_pack /
    wmz_layer_fused_bwd_pack produce stream by stream, and stamps the cache entries the forward / backward look up."""
    from world_modelz_amd import _cast, fused
    from world_modelz_amd.local_3d_attention import Local3dAttentionTransformer
    torch.manual_seed(6)
    tr = Local3dAttentionTransformer(data_shape=(2, 16, 16), dim=D, num_classes=32, extents=(1, 1, 1), depth=3, mlp_dim=M,
                                     dim_head=I, heads=1).cuda()
    with torch.no_grad():
        for p in tr.parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    layers = list(tr.layers)
    _cast.clear()
    ref = []
    for head, tail in [(None, layers[0])] + [(layers[l], layers[l + 1] if l + 1 < 3 else None) for l in range(3)]:
        ref.append([t.clone() for t in fused._layer_pack(head, tail)])
    for attn, ff in layers:
        ref.append([t.clone() for t in fused._layer_pack_bwd(attn, ff)])
    _cast.clear()
    ps = fused.PackSet(tr)
    ps.refresh()
    torch.cuda.synchronize()
    got = []
    for head, tail in [(None, layers[0])] + [(layers[l], layers[l + 1] if l + 1 < 3 else None) for l in range(3)]:
        got.append(fused._layer_pack(head, tail))          # cache hits: the PackSet's persistent buffers
    for attn, ff in layers:
        got.append(fused._layer_pack_bwd(attn, ff))
    assert any(g[0].data_ptr() == e[2][0].data_ptr() for g in got for e in ps.entries)
    for r, g in zip(ref, got):
        for a, b in zip(r, g):
            assert a.shape == b.shape and torch.equal(a, b)


def test_linear_wgrad_batch_vs_torch():
    """wmz_linear_wgrad_batch: several weight gradients of different shapes by one launch pair; accumulate vs overwrite,
    with and without a bias gradient, against fp32 torch on the same bf16 operands."""
    from world_modelz_amd import ops
    torch.manual_seed(4)
    n = 4096 + 64                                         # not a multiple of the slice length
    shapes = [(256, 256, True, False), (256, 128, False, True), (128, 256, True, True), (64, 512, False, False)]
    probs, refs = [], []
    for N, K, bias, overwrite in shapes:
        dc = torch.randn(n, N, device='cuda').bfloat16()
        a = torch.randn(n, K, device='cuda').bfloat16()
        dw0 = torch.randn(N, K, device='cuda')
        db0 = torch.randn(N, device='cuda') if bias else None
        g = dc.float().t() @ a.float()
        s = dc.float().sum(0)
        refs.append((g if overwrite else dw0 + g, None if db0 is None else (s if overwrite else db0 + s)))
        probs.append((dc, a, dw0.clone(), None if db0 is None else db0.clone(), overwrite))
    ops.linear_wgrad_batch(probs)
    torch.cuda.synchronize()
    for (dc, a, dw, db, ov), (rw, rb) in zip(probs, refs):
        assert rel(dw, rw) < 2e-5, (dw.shape, rel(dw, rw))
        if db is not None:
            assert rel(db, rb) < 2e-5


def test_linear_wgrad_batch_reads_the_tiled_stream():
    """a_tiled: A in the fused path's stream layout (per 32-row tile [16 chunks][2 halves][32 rows][8 features]) gives the
    gradient of the row-major A to the bit; an ordinary problem rides in the same launch."""
    from world_modelz_amd import ops
    torch.manual_seed(8)
    n = 2048 + 96
    a = torch.randn(n, 256, device='cuda').bfloat16()
    dc = torch.randn(n, 128, device='cuda').bfloat16()
    # row-major [n, 256] -> tiles: feature f = 128 h + 8 s + j of row 32 T + t  ->  [T][s][h][t][j]
    a_t = a.view(n // 32, 32, 2, 16, 8).permute(0, 3, 2, 1, 4).contiguous()
    g_rm, g_t = torch.empty(128, 256, device='cuda'), torch.empty(128, 256, device='cuda')
    other = torch.zeros(64, 128, device='cuda')
    dc2, a2 = torch.randn(n, 64, device='cuda').bfloat16(), torch.randn(n, 128, device='cuda').bfloat16()
    ops.linear_wgrad_batch([(dc, a, g_rm, None, True)])
    ops.linear_wgrad_batch([(dc, a_t.view(n, 256), g_t, None, True, True), (dc2, a2, other, None, False)])
    torch.cuda.synchronize()
    assert torch.equal(g_rm, g_t)
    assert rel(g_t, dc.float().t() @ a.float()) < 2e-5
    assert rel(other, dc2.float().t() @ a2.float()) < 2e-5


@pytest.mark.parametrize('shape,classes,skew', [((1, 2, 3, 16), 7, 'uniform'), ((2, 3, 5, 16), 40, 'dominant'),
                                                ((8, 32, 16, 16), 1025, 'last_frame_mask'), ((2, 3, 16, 16), 65, 'uniform')])
def test_embed_backward_sorted_gather_vs_torch(shape, classes, skew):
    """wmz_embed_pos3d_bwd_sorted (counting sort by class + gather) against index_add in fp32: the four tables, accumulation
    into non-zero tables, two calls in a row (the counters must be back at zero), class distributions with a dominant class
    (half of a frame masked; one code carrying most tokens), token counts that are not a multiple of the 64-entry wave tile,
    out-of-range ids clamped like the forward does.  (The cases share ONE workspace, class counts going up and down: the
    counters of a call must not sit in an earlier call's scratch.)"""
    from world_modelz_amd import ops
    torch.manual_seed(6)
    B, S, H, W = shape
    z = torch.randint(0, classes, shape, device='cuda')
    if skew == 'last_frame_mask':
        m = torch.rand(B, H, W, device='cuda') < 0.5
        z[:, -1][m] = classes - 1
    elif skew == 'dominant':
        z[torch.rand(shape, device='cuda') < 0.7] = 3
        z[0, 0, 0, 0], z[0, 0, 0, 1] = -5, classes + 9                   # clamped to 0 / classes - 1
    dx = torch.randn(*shape, 256, device='cuda').bfloat16()
    tabs0 = [torch.randn(n, 256, device='cuda') for n in (classes, S, H, W)]
    tabs = [t.clone() for t in tabs0]
    ops.embed_pos3d_bwd(z, dx, tabs)
    ops.embed_pos3d_bwd(z, dx, tabs)                                     # second call: counters re-zeroed by the first
    torch.cuda.synchronize()
    f = dx.float().reshape(-1, 256).double()
    zc = z.clamp(0, classes - 1).reshape(-1)
    ref = [t.double().clone() for t in tabs0]
    ref[0].index_add_(0, zc, 2 * f)
    g5 = 2 * dx.double()
    ref[1] += g5.sum((0, 2, 3))
    ref[2] += g5.sum((0, 1, 3))
    ref[3] += g5.sum((0, 1, 2))
    for name, a, b in zip(('emb', 'pos_s', 'pos_h', 'pos_w'), tabs, ref):
        err = float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
        assert err < 2e-6, (name, err)
