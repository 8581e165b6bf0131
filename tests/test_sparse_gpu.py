"""GPU parity of the config-5 path (minecraft/sparse_diffusion.py + transformer.py drop-ins) against the reference
capture and the oracle's autograd."""
import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu

from oracle import denoiser as oden          # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def wmz():
    assert torch.cuda.is_available()
    from world_modelz_amd import config, sparse_diffusion
    return dict(config=config, sd=sparse_diffusion)


def test_sparse_model_vs_golden_and_grads(wmz):
    g = load_golden('sparse_tiny')
    sd = sub(g, 'sd/')
    shape = tuple(int(e) for e in g['shape'])
    heads = int(g['heads'])
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=32, num_classes=40, depth=2, dim_head=16, mlp_dim=48, heads=heads)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x, idx = g['x'].cuda(), g['indices'].cuda()
    with wmz['config'].compute_dtype(torch.float32):
        with torch.no_grad():
            y = m(x, idx)
        assert y.shape == g['logits'].shape and rel(y, g['logits']) < 1e-5
        # gradients against torch.autograd over the oracle
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = oden.sparse_denoiser_forward(leaves, g['x'], g['indices'], shape, heads)
        w = torch.randn_like(ref)
        (ref * w).sum().backward()
        out = m(x, idx)
        (out * w.cuda()).sum().backward()
    for n, p in m.named_parameters():
        assert rel(p.grad, leaves[n].grad) < 5e-5, n


def test_sparse_model_dropout_in_training(wmz):
    """dropout > 0 in config 5's blocks (minecraft/transformer.py:21-31, :44-62: on the attention PROBABILITIES, behind to_out and
    inside the feed-forward; sparse_diffusion.py:76 default 0): used to raise in training.  Eval agrees with the dropout-0 model;
    with every mask drawn as keep (p -> the torch path but dropout 1e-9) the un-fused attention path reproduces the fused one;
    training draws differ and their mean approaches eval; gradients reach every parameter."""
    torch.manual_seed(5)
    kw = dict(shape=(4, 8, 8), dim=64, num_classes=40, depth=2, dim_head=32, mlp_dim=96, heads=2)
    m0 = wmz['sd'].VqSparseDiffusionModel(**kw).cuda()
    md = wmz['sd'].VqSparseDiffusionModel(dropout=0.25, **kw).cuda()
    me = wmz['sd'].VqSparseDiffusionModel(dropout=1e-9, **kw).cuda()
    md.load_state_dict(m0.state_dict())
    me.load_state_dict(m0.state_dict())
    x = torch.randint(0, 41, (3, 48), device='cuda')
    idx = torch.stack([torch.randperm(256, device='cuda')[:48] for _ in range(3)])
    with wmz['config'].compute_dtype(torch.float32):
        m0.eval(); md.eval()
        with torch.no_grad():
            y0 = m0(x, idx)
            assert torch.allclose(md(x, idx), y0, rtol=1e-5, atol=1e-6)
            me.train()
            assert rel(me(x, idx), y0) < 1e-5                  # the torch-composed attention == the HIP attention block
        md.train()
        with torch.no_grad():
            ys = torch.stack([md(x, idx) for _ in range(48)])
        assert not torch.equal(ys[0], ys[1])
        err, spread = rel(ys.mean(0), y0), rel(ys[0], y0)
        assert spread > 0.05 and err < 0.5 * spread, (spread, err)
        md.zero_grad()
        md(x, idx).square().mean().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0 for p in md.parameters())


@pytest.mark.parametrize('n', [512, 100])
def test_sparse_model_default_width_bf16(wmz, n):
    """dim 512, 4 heads x 128, mlp 1024 (sparse_diffusion.py:233-257), n = 512 tokens (16-wide fast path) and a
    ragged n = 100 (general kernel), bf16 vs the fp32 oracle."""
    torch.manual_seed(3)
    shape = (8, 16, 16)
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=512, num_classes=256, depth=2, dim_head=128, mlp_dim=1024, heads=4)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randint(0, 257, (2, n))
    idx = torch.stack([torch.randperm(8 * 256)[:n] for _ in range(2)])
    ref = oden.sparse_denoiser_forward(sd, x, idx, shape, 4)
    m = m.cuda()
    with torch.no_grad():
        with wmz['config'].compute_dtype(torch.float32):
            y32 = m(x.cuda(), idx.cuda())
        with wmz['config'].compute_dtype(torch.bfloat16):
            y16 = m(x.cuda(), idx.cuda())
    assert rel(y32, ref) < 1e-5
    assert rel(y16, ref) < 2e-2


def test_position_samplers(wmz):
    sdm = wmz['sd']
    p = sdm.sample_flat_positions(3, 100, 6, 4, 4, 'cuda')
    assert p.shape == (3, 100) and int(p.min()) >= 0 and int(p.max()) < 96
    t = torch.tensor([0.0, 0.5, 1.0])
    q = sdm.sample_time_dependent(3, 32, 8, 4, 4, t, 'cuda', o=torch.tensor([0.0, 0.5, 0.99]))
    assert q.shape == (3, 32)
    for b in range(3):
        assert len(set(q[b].tolist())) == 32                      # without replacement
    # t = 0: window = min_sample_window = 2 frames -> positions within 2*16 of the offset
    assert int(q[0].max()) - int(q[0].min()) < 2 * 16
    assert int(q.max()) < 8 * 16


def test_fused_context_draw_vs_reference_law(wmz):
    """wmz_sparse_draw_context (one launch for sparse_diffusion.py:44-72 sample_time_dependent + :437 gather + :440-449 corruption):
    the window is the reference's (first frame, frame count: its fp32 formula, restated here on the CPU), the positions are distinct
    and inside it, targets are the clip's tokens there, the corruption obeys the law at r = 0 / r = 1 / in between, the draw is a
    function of (seed, rank, counter) and changes with each of them, and over many draws every position of the window is equally
    likely in every output slot (uniform without replacement, random order)."""
    import math
    from world_modelz_amd import train
    torch.manual_seed(3)
    S, H, W, n, C = 8, 4, 4, 32, 50
    HW = H * W
    B = 5
    z = torch.randint(0, C, (B, S, H, W), device='cuda')
    r = torch.tensor([0.0, 0.3, 0.6, 1.0, 0.45])
    o = torch.tensor([0.0, 0.5, 0.99, 0.2, 0.7])
    ctr = torch.full((1,), 7, dtype=torch.int64, device='cuda')
    idx, tok, tgt = train.draw_sparse_context(z, r, n, (S, H, W), C, seed=11, rank=0, counter=ctr, o=o)
    need = math.ceil(n / HW)
    frames = torch.floor(need + r.clamp(0, 1) * (S - need + 1)).clamp(max=S - need)
    first = torch.floor(o.clamp(0, 1 - 1e-5) * (S - frames + 1))
    for b in range(B):
        lo, hi = int(first[b]) * HW, (int(first[b]) + int(frames[b])) * HW
        v = idx[b].tolist()
        assert len(set(v)) == n and min(v) >= lo and max(v) < hi, (b, lo, hi, min(v), max(v))
    assert torch.equal(tgt, torch.gather(z.reshape(B, -1), 1, idx))
    assert torch.equal(tok[0], tgt[0]) and bool((tok[3] == C).all())                      # r = 0: untouched; r = 1: all masked
    masked = float((tok[2] == C).float().mean())
    assert 0.3 < masked < 0.9, masked                                                       # r = 0.6 of 32 tokens
    again = train.draw_sparse_context(z, r, n, (S, H, W), C, seed=11, rank=0, counter=ctr, o=o)
    assert all(torch.equal(a, b_) for a, b_ in zip((idx, tok, tgt), again))               # a function of (seed, rank, counter)
    for kw in (dict(seed=12, rank=0), dict(seed=11, rank=1)):
        other = train.draw_sparse_context(z, r, n, (S, H, W), C, counter=ctr, o=o, **kw)
        assert not torch.equal(other[0], idx)
    ctr += 1
    other = train.draw_sparse_context(z, r, n, (S, H, W), C, seed=11, rank=0, counter=ctr, o=o)
    assert not torch.equal(other[0], idx)
    # uniformity: one clip, fixed window (r = 1 -> frames = S - need = 6 of 8, o = 0 -> first = 0): 96 positions, 32 drawn per call
    reps = 1500
    zz = z[:1].expand(reps, S, H, W).contiguous()
    draws = train.draw_sparse_context(zz, torch.ones(reps), n, (S, H, W), C, seed=5, rank=0, o=torch.zeros(reps))[0]   # [reps, n]
    Wn = 6 * HW
    assert int(draws.max()) < Wn
    counts = torch.bincount(draws.reshape(-1), minlength=Wn).float()                      # each position: reps * n / Wn = 500 expected
    exp = reps * n / Wn
    assert float((counts - exp).abs().max()) < 6 * math.sqrt(exp), (counts.min(), counts.max(), exp)
    slot0 = torch.bincount(draws[:, 0], minlength=Wn).float()                             # the first slot alone: reps / Wn = 15.6 expected
    assert float(slot0.max()) < 45 and float((slot0 == 0).float().mean()) < 0.01         # (random order: no position favoured up front)
    assert abs(float(draws[:, 0].float().mean()) - (Wn - 1) / 2) < 4 * (Wn / math.sqrt(12 * reps))
    # the window placement drawn in the kernel (o = None) covers every admissible first frame
    firsts = train.draw_sparse_context(zz, torch.zeros(reps), n, (S, H, W), C, seed=6, rank=0)[0].min(dim=1).values // HW
    assert set(firsts.tolist()) == set(range(S - need + 1)), set(firsts.tolist())


def test_fused_context_config5_shape_and_unsupported_grids(wmz):
    """Config 5's own shape (64 x 16 x 16 = 16 384 positions, 512 context tokens: 128 KB of LDS keys), and the shapes the kernel
    declines (more than 65 536 positions; a grid too short for the narrowest window): there the trainer draws with the torch ops."""
    from world_modelz_amd import train
    from world_modelz_amd import _lib as L
    S, H, W, n, C, B = 64, 16, 16, 512, 8192, 6
    z = torch.randint(0, C, (B, S, H, W), device='cuda')
    r = torch.tensor([0.0, 0.2, 0.4, 0.6, 0.8, 1.0])
    idx, tok, tgt = train.draw_sparse_context(z, r, n, (S, H, W), C, seed=1, rank=0)
    for b in range(B):
        assert idx[b].unique().numel() == n
    assert int(idx.min()) >= 0 and int(idx.max()) < S * H * W
    assert torch.equal(tgt, torch.gather(z.reshape(B, -1), 1, idx))
    span = (idx.max(dim=1).values - idx.min(dim=1).values).tolist()
    assert span[0] < 2 * 256 + 256 and span[-1] > 40 * 256                                 # narrow window at r = 0, wide at r = 1
    assert L.lib().wmz_sparse_draw_context_supported(64, 256, 512) == 1
    assert L.lib().wmz_sparse_draw_context_supported(257, 256, 512) == 0                   # > 65 536 positions
    assert L.lib().wmz_sparse_draw_context_supported(3, 16, 32) == 0                       # 2 * ceil(n / HW) > S
    with pytest.raises(L.WmzError):
        train.draw_sparse_context(torch.zeros((1, 257, 16, 16), dtype=torch.int64, device='cuda'), torch.ones(1), 512, (257, 16, 16), C)
    # a grid four times config 5's (65 536 positions: 32 keys a histogram bin)
    zb = torch.randint(0, C, (2, 256, 16, 16), device='cuda')
    ib, _, tb = train.draw_sparse_context(zb, torch.tensor([1.0, 0.5]), n, (256, 16, 16), C, seed=2, rank=0)
    assert ib[0].unique().numel() == n and ib[1].unique().numel() == n and torch.equal(tb, torch.gather(zb.reshape(2, -1), 1, ib))


def test_sparse_training_step_vs_oracle_autograd(wmz):
    """SparseDenoiserTrainer (minecraft/sparse_diffusion.py:398-467): one fp32 step on the reference capture's model -- the
    chunked linear + cross-entropy (no [R, C] logits in memory) and the dense-attention backward against torch.autograd over
    the oracle, then the AdamW update against torch.optim.AdamW; and the corruption law on a [B, n] token matrix."""
    from world_modelz_amd import train
    g = load_golden('sparse_tiny')
    sd = sub(g, 'sd/')
    shape = tuple(int(e) for e in g['shape'])
    heads, C = int(g['heads']), 40
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=32, num_classes=C, depth=2, dim_head=16, mlp_dim=48, heads=heads)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x, idx = g['x'].clamp(max=C - 1), g['indices']
    target = (x * 7 + 3) % C
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = oden.sparse_denoiser_forward(leaves, x, idx, shape, heads)
    loss_rows = torch.nn.functional.cross_entropy(logits.reshape(-1, C), target.reshape(-1), reduction='none')
    loss_rows.mean().backward()
    with wmz['config'].compute_dtype(torch.float32):
        tr = train.SparseDenoiserTrainer(m, C, num_context=16, lr=3e-3, warmup=0, distributed=False)
        tr.arena.zero_grad()
        per, mean = tr.forward_backward(x.cuda(), idx.cuda(), target.cuda())
        assert abs(float(mean) - float(loss_rows.mean())) < 1e-5
        assert torch.allclose(per.cpu(), loss_rows.view(3, -1).mean(1).detach(), rtol=1e-5)
        for n, p in m.named_parameters():
            assert rel(p.grad, leaves[n].grad) < 5e-5, n
        params = [leaves[n] for n, _ in m.named_parameters()]
        opt = torch.optim.AdamW(params, lr=3e-3, betas=(0.9, 0.999), weight_decay=1e-7)
        opt.step()
        tr.optimizer_step(lr=3e-3)
    for n, p in m.named_parameters():
        assert torch.allclose(p.detach().cpu(), leaves[n].detach(), rtol=2e-5, atol=2e-7), n
    # corruption on [B, n]: r = 0 keeps every token, r = 1 masks every token
    tok = torch.randint(0, C, (4, 64), device='cuda')
    out0, tgt0 = train.corrupt_tokens(tok, torch.zeros(4), C)
    out1, _ = train.corrupt_tokens(tok, torch.ones(4), C)
    assert torch.equal(out0, tok) and torch.equal(tgt0, tok) and bool((out1 == C).all())


def test_sparse_step_config5_size_properties(wmz):
    """BASELINE configs[4] at its real size (sparse_diffusion.py:233-257 with S and the codebook of the config: 64-frame
    clips of 16x16 latents, codebook 8192, 512 context tokens, dim 512, 4 heads x 128, depth 8, mlp 1024), 6 clips (the per-GPU
    share of batch 48 on 8 GPUs), bf16.  The fp32 oracle is too slow here; properties instead:
      * a clip's per-sample loss does not depend on its batch neighbours; gradients are additive over clips;
      * chunked linear + cross-entropy == the dense logits + CE (loss rows, dx, dW) on the same activations;
      * a training step runs, keeps everything finite and lowers the loss on a repeated batch."""
    from world_modelz_amd import train
    torch.manual_seed(50)
    shape, C, n, B = (64, 16, 16), 8192, 512, 6
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=512, num_classes=C, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
    z = torch.randint(0, C, (B,) + shape, device='cuda')
    with wmz['config'].compute_dtype(torch.bfloat16):
        tr = train.SparseDenoiserTrainer(m, C, num_context=n, lr=3e-4, warmup=0, distributed=False)
        r = torch.full((B,), 0.5)
        idx = tr.sample_positions(B, r, z.device)
        assert idx.shape == (B, n) and int(idx.max()) < 64 * 256
        for b in range(B):
            assert idx[b].unique().numel() == n
        tok = torch.gather(z.reshape(B, -1), 1, idx)
        target = tok.clone()
        tok = torch.where(torch.rand(B, n, device='cuda') < 0.5, torch.full_like(tok, C), tok)

        def grads(sel):
            tr.arena.zero_grad()
            per, mean = tr.forward_backward(tok[sel], idx[sel], target[sel])
            return per.clone(), tr.arena.flat_grad.clone()
        per_all, g_all = grads(slice(0, B))
        per_a, g_a = grads(slice(0, 3))
        per_b, g_b = grads(slice(3, B))
        assert torch.isfinite(per_all).all() and torch.isfinite(g_all).all()
        assert torch.equal(per_a, per_all[:3]) and torch.equal(per_b, per_all[3:])
        add = float((0.5 * (g_a + g_b) - g_all).norm() / g_all.norm())
        print(f'[config-5 step] clip additivity of the gradient: {add:.2e}')
        assert add < 2e-3
        # chunked linear-CE == dense logits + CE
        h = torch.randn(1024 + 300, 512, device='cuda').bfloat16().requires_grad_(True)
        tg = torch.randint(0, C, (h.shape[0],), device='cuda')
        w, bb = m.logit_proj.weight, m.logit_proj.bias
        for p in (w, bb):                                  # take the non-arena path: fresh gradients through autograd
            p.__dict__.pop('_wmz_grad', None)
            p.grad = None
        mean_c, rows_c = train.linear_cross_entropy(h, w, bb, tg, chunk=512)
        mean_c.backward()
        dh_c, dw_c, db_c = h.grad.clone(), w.grad.clone(), bb.grad.clone()
        h.grad = None; w.grad = None; bb.grad = None
        logits = h.float() @ w.t() + bb
        rows_d = torch.nn.functional.cross_entropy(logits, tg, reduction='none')
        rows_d.mean().backward()
        assert rel(rows_c, rows_d) < 2e-3 and rel(dh_c, h.grad) < 2e-2 and rel(dw_c, w.grad) < 2e-2 and rel(db_c, bb.grad) < 2e-2
    del tr
    with wmz['config'].compute_dtype(torch.bfloat16):
        tr = train.SparseDenoiserTrainer(m, C, num_context=n, lr=3e-4, warmup=0, distributed=False)
        losses = [tr.train_step(z, r=torch.full((B,), 0.3), indices=idx)[0] for _ in range(6)]
    print('[config-5 step] loss over 6 steps on one batch:', [f'{v:.3f}' for v in losses])
    assert all(l == l for l in losses) and losses[-1] < losses[0]


def test_sparse_graphed_training_step(wmz):
    """SparseDenoiserTrainer.enable_graph (config 5's step as ONE hipGraph: position sampling from torch's capture-aware device
    RNG, gather, corruption from the device-counted Philox stream, forward, chunked linear + CE, backward, AdamW).  Every replay
    draws fresh positions and a fresh corruption; training on one batch reduces the loss like the eager trainer does; with the
    positions given (`indices=`) the trainer still takes the eager path."""
    from world_modelz_amd import train
    from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
    C, B, n = 512, 3, 512
    def make():
        torch.manual_seed(41)
        return VqSparseDiffusionModel(shape=(16, 16, 16), dim=512, num_classes=C, depth=2, dim_head=128, mlp_dim=1024, heads=4).cuda()
    torch.manual_seed(42)
    z = torch.randint(0, C, (B, 16, 16, 16), device='cuda')
    r = torch.full((B,), 0.3)
    with wmz['config'].compute_dtype(torch.bfloat16):
        mg, me = make(), make()
        tg = train.SparseDenoiserTrainer(mg, C, num_context=n, lr=3e-4, warmup=0, distributed=False)
        te = train.SparseDenoiserTrainer(me, C, num_context=n, lr=3e-4, warmup=0, distributed=False)
        p0 = tg.arena.flat_param.clone()
        tg.enable_graph(z)
        assert torch.equal(tg.arena.flat_param, p0) and tg.step_count == 0      # warm-up updates rolled back
        lg = [tg.train_step(z, r=r)[0] for _ in range(12)]
        le = [te.train_step(z, r=r)[0] for _ in range(12)]
        assert tg._graph is not None and tg.step_count == 12
        print('[config-5 graphed] loss', [f'{v:.3f}' for v in lg], 'eager', [f'{v:.3f}' for v in le])
        assert all(v == v for v in lg) and lg[-1] < lg[0]
        assert abs(lg[-1] - le[-1]) < 0.15 * le[0]              # same law, different random streams
        assert len({round(v, 6) for v in lg[:4]}) == 4           # fresh positions / corruption on every replay
        idx = tg.sample_positions(B, r, z.device)
        out = tg.train_step(z, r=r, indices=idx)                 # injected positions: eager path
        assert out[0] == out[0] and tg.step_count == 13
        # the captured step draws its context by ONE launch (wmz_sparse_draw_context); with the knob off the torch ops of
        # sparse_diffusion.py do (the same law from torch's device generator): same training behaviour
        assert tg.fused_context(z)
        mt = make()
        tt = train.SparseDenoiserTrainer(mt, C, num_context=n, lr=3e-4, warmup=0, distributed=False)
        tt.use_fused_context = False
        tt.enable_graph(z)
        lt = [tt.train_step(z, r=r)[0] for _ in range(12)]
        print('[config-5 graphed] torch-op prologue loss', [f'{v:.3f}' for v in lt])
        assert lt[-1] < lt[0] and abs(lg[-1] - lt[-1]) < 0.15 * lt[0]


def test_sparse_model_dim_head_off_the_granule(wmz):
    """minecraft/transformer.py's Attention with a dim_head that is no multiple of 8 (3 heads of 20): logits and gradients against
    the fp32 oracle (the heads run zero-padded: transformer.py::_head_padded)."""
    from oracle import denoiser as oden
    torch.manual_seed(8)
    shape, C, heads, dh = (4, 4, 4), 30, 3, 20
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=48, num_classes=C, depth=2, dim_head=dh, mlp_dim=64, heads=heads)
    leaves = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = torch.randint(0, C + 1, (2, 32))
    idx = torch.stack([torch.randperm(64)[:32] for _ in range(2)])
    ref = oden.sparse_denoiser_forward(leaves, x, idx, shape, heads)
    ref.square().mean().backward()
    m = m.cuda()
    with wmz['config'].compute_dtype(torch.float32):
        y = m(x.cuda(), idx.cuda())
        y.square().mean().backward()
    assert rel(y, ref) < 1e-5
    worst = max((rel(p.grad, leaves[n].grad), n) for n, p in m.named_parameters() if leaves[n].grad is not None)
    print(f'sparse model, 3 heads of 20: logits {rel(y, ref):.2e}, worst gradient {worst[0]:.2e} ({worst[1]})')
    assert worst[0] < 3e-4, worst


def test_categorical_scatter_vs_softmax(wmz):
    """wmz_categorical_scatter (sparse_diffusion.py:190-197: softmax -> multinomial -> scatter_): draws follow softmax(logits)
    (empirical frequencies over 4 000 calls against the probabilities), a dominating logit is always drawn (C = 8192, config 5's
    vocabulary, three passes over the row), the draw is a function of (seed, call), and only the rows' positions of the clips change."""
    sdm = wmz['sd']
    torch.manual_seed(2)
    B, n, C, G = 2, 8, 12, 40
    logits = (2.0 * torch.randn(B, n, C, device='cuda')).contiguous()
    idx = torch.stack([torch.randperm(G, device='cuda')[:n] for _ in range(B)]).contiguous()
    p = torch.softmax(logits, -1)
    counts = torch.zeros(B, n, C, device='cuda')
    base = torch.full((B, G), 99, dtype=torch.int64, device='cuda')
    reps = 4000
    for call in range(reps):
        z = base.clone()
        sdm.categorical_scatter(logits, idx, z, seed=7, call_id=call)
        counts.scatter_add_(2, torch.gather(z, 1, idx).unsqueeze(-1), torch.ones(B, n, 1, device='cuda'))
    untouched = torch.ones(B, G, dtype=torch.bool, device='cuda').scatter_(1, idx, False)
    assert bool((z[untouched] == 99).all()) and int(torch.gather(z, 1, idx).max()) < C
    err = (counts / reps - p).abs()
    assert float(err.max()) < 5 * 0.5 / (reps ** 0.5), float(err.max())                   # 5 sigma of a Bernoulli frequency
    z1, z2, z3 = base.clone(), base.clone(), base.clone()
    sdm.categorical_scatter(logits, idx, z1, seed=7, call_id=3)
    sdm.categorical_scatter(logits, idx, z2, seed=7, call_id=3)
    sdm.categorical_scatter(logits, idx, z3, seed=8, call_id=3)
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    # config 5's vocabulary: a dominating class per row is drawn with certainty, wherever it sits in the row
    C = 8192
    big = torch.randn(3, 16, C, device='cuda')
    win = torch.randint(0, C, (3, 16), device='cuda')
    win[0, 0], win[0, 1] = 0, C - 1
    big.scatter_(2, win.unsqueeze(-1), 60.0)
    pos = torch.stack([torch.randperm(G, device='cuda')[:16] for _ in range(3)]).contiguous()
    zz = torch.zeros(3, G, dtype=torch.int64, device='cuda')
    sdm.categorical_scatter(big.contiguous(), pos, zz, seed=1, call_id=1)
    assert torch.equal(torch.gather(zz, 1, pos), win)


def test_sparse_sampler_plumbing_and_model_run(wmz):
    """sample_clips (the token side of the reference's evaluate_model, sparse_diffusion.py:139-202).  With a stand-in model whose
    logits put all mass on `position mod C` the sampler's plumbing is exact: every position a context visited holds its own
    class, the rest are still masked, the result is a function of the seed; with the real model it runs in both sampling modes,
    leaves only valid tokens or masks, and evaluate_model returns decoded frames of the reference's shape."""
    from world_modelz_amd.train_vqae import VqAutoEncoder
    sdm = wmz['sd']
    S, H, W, C, n = 8, 4, 4, 24, 32
    G = S * H * W

    class Oracle(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.shape = (S, H, W)
            self.embedding = torch.nn.Embedding(C + 1, 8)
            self.seen_masked = []

        def forward(self, tokens, indices):
            self.seen_masked.append(float((tokens == C).float().mean()))
            return 50.0 * torch.nn.functional.one_hot(indices % C, C).float()

    stub = Oracle().cuda()
    g = torch.Generator().manual_seed(3)
    z = sdm.sample_clips(stub, 3, C, 'neighbors', num_context=n, num_eval_iterations=5, generator=g, seed=11, use_graph=False)   # (the stand-in reads back)
    flat = z.view(3, -1)
    pos = torch.arange(G, device='cuda').expand(3, -1)
    visited = flat != C
    assert bool((flat[visited] == (pos % C)[visited]).all()) and float(visited.float().mean()) > 0.9
    per_iter = G // n + 1
    # iteration 0 sees only masks; the last one masks nothing itself (what is still masked there was never visited)
    assert stub.seen_masked[0] == 1.0 and max(stub.seen_masked[-per_iter:]) < 0.35, stub.seen_masked[-per_iter:]
    z2 = sdm.sample_clips(stub, 3, C, 'neighbors', num_context=n, num_eval_iterations=5, generator=torch.Generator().manual_seed(3), seed=11,
                          use_graph=False)
    assert torch.equal(z, z2)
    zu = sdm.sample_clips(stub, 2, C, 'uniform', num_context=n, num_eval_iterations=3, generator=torch.Generator().manual_seed(3), seed=11)
    fu = zu.view(2, -1)
    assert bool((fu == (torch.arange(G, device='cuda') % C)).all())                        # a permutation's slices cover the grid
    # the real model, both modes
    torch.manual_seed(4)
    m = sdm.VqSparseDiffusionModel(shape=(S, H, W), dim=64, num_classes=C, depth=2, dim_head=32, mlp_dim=96, heads=2).cuda()
    with wmz['config'].compute_dtype(torch.bfloat16):
        for mode in ('neighbors', 'uniform'):
            zz = sdm.sample_clips(m, 2, C, mode, num_context=n, num_eval_iterations=3, seed=5, generator=torch.Generator().manual_seed(1))
            assert zz.shape == (2, S, H, W) and int(zz.min()) >= 0 and int(zz.max()) <= C
            if mode == 'neighbors':                    # the captured sub-step (default) draws what the eager launches draw
                ze = sdm.sample_clips(m, 2, C, mode, num_context=n, num_eval_iterations=3, seed=5, generator=torch.Generator().manual_seed(1),
                                      use_graph=False)
                assert torch.equal(zz, ze)
        ae = VqAutoEncoder(embedding_dim=16, num_embeddings=C, downscale_steps=2, hidden_planes=16).cuda().eval()
        frames = sdm.evaluate_model('cuda', 2, m, ae, (S, H, W), 'neighbors', num_context=n, num_eval_iterations=2)
    assert frames.shape == (2, S, 3, 4 * H, 4 * W) and bool(torch.isfinite(frames).all())
