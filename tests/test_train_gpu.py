"""GPU: the training-step driver (world_modelz_amd/train.py) against the reference captures."""
import math

import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def wmz():
    assert torch.cuda.is_available()
    from world_modelz_amd import config, main, train
    return dict(config=config, main=main, train=train)


def _tiny_model(wmz, g):
    sd0 = sub(g, 'sd0/')
    ext = tuple(int(e) for e in g['extents'])
    C = g['logits'].shape[-1]
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 4, 4), dim=16, num_classes=C, extents=ext, depth=2, dim_head=8,
                                          mlp_dim=24, heads=int(g['heads']))
    m.load_state_dict(sd0, strict=True)
    return m.cuda(), C


def test_one_adamw_step_matches_reference(wmz):
    """forward/backward + flat-arena grad-norm + AdamW kernel == torch.optim.AdamW on the reference model."""
    g = load_golden('step_tiny')
    m, C = _tiny_model(wmz, g)
    with wmz['config'].compute_dtype(torch.float32):
        tr = wmz['train'].DenoiserTrainer(m, C, lr=float(g['adamw_lr']), warmup=0, max_steps=1000, distributed=False)
        tr.arena.zero_grad()
        per_sample, mean = tr.forward_backward(g['corrupted'].cuda(), g['target'].cuda())
        sq = tr.optimizer_step(lr=float(g['adamw_lr']))
    assert torch.allclose(per_sample.cpu(), g['per_sample_loss'], rtol=1e-5)
    assert math.isclose(math.sqrt(float(sq)), float(g['grad_norm']), rel_tol=1e-4)
    sd1 = sub(g, 'sd1/')
    for n, p in m.named_parameters():
        assert torch.allclose(p.detach().cpu(), sd1[n], rtol=2e-5, atol=2e-7), n
    # the operand caches were invalidated: a second forward sees the updated weights
    with wmz['config'].compute_dtype(torch.float32), torch.no_grad():
        y2 = m(g['corrupted'].cuda())
    assert not torch.allclose(y2.cpu(), g['logits'], rtol=1e-4)


def test_corruption_matches_reference_law(wmz):
    """Closed-form corruption == multinomial(lerp(one_hot, 1/C, 0.1 r)) + mask (main.py:246-259) in distribution."""
    torch.manual_seed(0)
    C, B, HW = 8, 2, 4096
    z = torch.full((B, 2, 64, 64), 3, device='cuda')
    r = torch.tensor([0.6, 1.0])
    zc, target = wmz['train'].corrupt_last_frame(z, r, C)
    assert torch.equal(target, z[:, -1]) and torch.equal(zc[:, 0], z[:, 0])
    for b in range(B):
        cnt = torch.bincount(zc[b, -1].reshape(-1).cpu(), minlength=C + 1).float() / HW
        a, rb = 0.1 * float(r[b]), float(r[b])
        expect = torch.full((C + 1,), (1 - rb) * a / C)
        expect[3] = (1 - rb) * (1 - a + a / C)
        expect[C] = rb
        assert torch.allclose(cnt, expect, atol=0.02), (cnt, expect)


def test_corruption_is_keyed_by_the_data_parallel_rank(wmz):
    """Ranks started from the same torch seed must not apply one mask / redraw pattern to their different clips: the
    Philox stream id carries the rank.  Same (seed, rank, call) -> same draw; another rank -> another draw."""
    tr = wmz['train']
    z = torch.randint(0, 16, (2, 3, 32, 32), device='cuda')
    r = torch.tensor([0.5, 0.5])
    outs = {}
    for rank in (0, 1, 0):
        tr._corrupt_calls = 41                                   # the same per-process call counter on every "rank"
        zc, _ = tr.corrupt_last_frame(z, r, 16, seed=1234, rank=rank)
        outs.setdefault(rank, []).append(zc[:, -1].clone())
    assert torch.equal(outs[0][0], outs[0][1])
    differ = float((outs[0][0] != outs[1][0]).float().mean())
    assert differ > 0.3, differ                                  # ~half the positions are masked, independently per rank


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gradient_accumulation_equals_the_big_batch(wmz, dtype):
    """main.py:221, :274-280: gradients accumulate over `accumulation_steps` micro-batches, each micro-loss scaled by
    1/acc_steps.  Two micro-batches of 2 clips must produce the gradient (and step) of one batch of 4 clips -- through the
    fused stack, its last-plane-only backward and the fused linear + cross-entropy in bf16, the op-by-op path in fp32."""
    tr = wmz['train']

    def make():
        torch.manual_seed(31)
        return wmz['main'].VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=256, num_classes=64, extents=(1, 1, 1), depth=2,
                                                 dim_head=128, mlp_dim=256, heads=1).cuda()
    z = torch.randint(0, 64, (4, 3, 16, 16), device='cuda')
    tgt = z[:, -1].clone()
    with wmz['config'].compute_dtype(dtype):
        ma, mb = make(), make()
        ta = tr.DenoiserTrainer(ma, 64, lr=1e-3, warmup=0, distributed=False, accumulation_steps=2)
        tb = tr.DenoiserTrainer(mb, 64, lr=1e-3, warmup=0, distributed=False)
        ta.arena.zero_grad()
        tb.arena.zero_grad()
        la = 0.0
        for i in (0, 2):
            _, m_ = ta.forward_backward(z[i:i + 2], tgt[i:i + 2], 0.5)
            la += 0.5 * float(m_)
        _, mb_ = tb.forward_backward(z, tgt, 1.0)
        lb = float(mb_)
    ga, gb = ta.arena.flat_grad, tb.arena.flat_grad
    err = float((ga - gb).norm() / gb.norm())
    print(f'[accumulation {dtype}] loss {la:.6f} vs {lb:.6f}, gradient rel {err:.2e}')
    assert abs(la - lb) < (1e-5 if dtype == torch.float32 else 2e-3) * max(1.0, abs(lb)), (la, lb)
    assert err < (1e-5 if dtype == torch.float32 else 6e-3), err


def test_training_reduces_loss(wmz):
    """A few full train_step() calls (corrupt -> fwd/bwd -> grad-norm -> AdamW) on a learnable toy task, bf16."""
    torch.manual_seed(1)
    C = 16
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 8, 8), dim=64, num_classes=C, extents=(1, 1, 1), depth=2, dim_head=32,
                                          mlp_dim=64, heads=2).cuda()
    tr = wmz['train'].DenoiserTrainer(m, C, lr=3e-3, warmup=1, max_steps=1000, distributed=False)
    z = torch.randint(0, C, (1, 1, 8, 8), device='cuda').expand(8, 3, 8, 8).contiguous()   # same frame repeated
    losses = []
    with wmz['config'].compute_dtype(torch.bfloat16):
        for _ in range(30):
            loss, gn = tr.train_step(z, r=torch.full((8,), 0.9))
            assert math.isfinite(loss) and math.isfinite(gn)
            losses.append(loss)
    assert losses[-1] < 0.5 * losses[0], losses


def test_sampler_loop_runs_and_unmasks(wmz):
    """Iterative-unmasking sampler (main.py:50-117 counterpart): shapes, value ranges, frame shift, determinism under a
    seeded generator on both paths -- the graphed one (draw + re-mask fused into the graph, in-kernel Philox keyed by the
    generator's seed) and the eager torch one (device RNG; with injected uniforms it is the parity path of the next test)."""
    from world_modelz_amd import sample
    torch.manual_seed(4)
    C = 16
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 8, 8), dim=64, num_classes=C, extents=(1, 1, 1), depth=2, dim_head=32,
                                          mlp_dim=64, heads=2).cuda().eval()
    z = torch.randint(0, C, (2, 3, 8, 8), device='cuda')
    outs = {}
    for use_graph in (True, False):
        gen = torch.Generator(device='cuda').manual_seed(9)
        frames, zf = sample.sample_frames(m, z, C, num_frames=2, num_eval_iterations=4, sample_topk=5, generator=gen,
                                          use_graph=use_graph)
        assert len(frames) == 2 and frames[0].shape == (2, 8, 8)
        for f in frames:
            assert int(f.min()) >= 0 and int(f.max()) < C           # last iteration: alpha = 1, nothing stays masked
        assert torch.equal(zf[:, 0], frames[0]) and torch.equal(zf[:, 1], frames[1])   # [c0,c1,L] -> [g1,g2,g2]
        outs[use_graph] = frames
        gen = torch.Generator(device='cuda').manual_seed(9)
        again, _ = sample.sample_frames(m, z, C, num_frames=2, num_eval_iterations=4, sample_topk=5, generator=gen,
                                        use_graph=use_graph)
        assert all(torch.equal(a, b) for a, b in zip(frames, again))
    lg = torch.randn(6, C, device='cuda')
    tk = sample.top_k_logits(lg, 3)
    assert int(torch.isfinite(tk).sum()) == 18


def test_sampler_session_is_reused_and_follows_the_weights(wmz):
    """The fused sampler keeps its captured step (forward + next draw + counter) with the model between calls (round 4: a call
    used to pay the capture again, ~4 ms against ~11 ms for a frame of 30 iterations): the second call replays the first call's
    graph, a reseeded generator reproduces a call on it, another generator state gives other tokens, and after the weights moved
    the SAME session re-captures (GraphedForward's stamp) and samples from the new model."""
    from world_modelz_amd import sample
    torch.manual_seed(6)
    C = 32
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 8, 8), dim=64, num_classes=C, extents=(1, 1, 1), depth=2, dim_head=32,
                                          mlp_dim=64, heads=2).cuda().eval()
    z = torch.randint(0, C, (2, 3, 8, 8), device='cuda')
    def run(seed):
        return sample.sample_frames(m, z, C, num_frames=2, num_eval_iterations=5, sample_topk=8, generator=torch.Generator().manual_seed(seed))[0]
    a = run(3)
    ses = next(iter(m._wmz_sampler_sessions.values()))
    g0 = ses.fwd.graph
    b, c = run(3), run(4)
    assert len(m._wmz_sampler_sessions) == 1 and ses.fwd.graph is g0 and ses.fwd.recaptures == 0
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and any(not torch.equal(x, y) for x, y in zip(a, c))
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(-1.0)                                   # (every logit flips sign: the draws change)
    d = run(3)
    assert ses.fwd.recaptures == 1 and any(not torch.equal(x, y) for x, y in zip(a, d))
    # a copy of the model starts without the sessions (they hold a hipGraph: not copyable), the original keeps its own
    import copy
    import pickle
    m2 = copy.deepcopy(m)
    assert len(m2._wmz_sampler_sessions) == 0 and len(m._wmz_sampler_sessions) == 1
    assert len(pickle.loads(pickle.dumps(m._wmz_sampler_sessions))) == 0
    # ... and it equals a fresh session on the moved weights
    m._wmz_sampler_sessions.clear()
    e = run(3)
    assert all(torch.equal(x, y) for x, y in zip(d, e))


@pytest.mark.parametrize('name', ['sampler_tiny', 'sampler_tiny_topk'])
@pytest.mark.parametrize('use_graph', [False, True])
def test_sampler_loop_token_for_token_vs_reference(wmz, name, use_graph):
    """evaluate_model (main.py:50-117) captured from the reference with injected uniforms: the GPU sampler must feed the
    model the same last frame in each of the 2 x 30 iterations and generate the same frames, token for token; the decoded
    images follow through the (train-mode BatchNorm, quirk Q3) VQ-AE decoder."""
    from world_modelz_amd import sample
    from world_modelz_amd.train_vqae import VqAutoEncoder
    g = load_golden(name)
    C = int(g['num_embeddings'])
    sd = sub(g, 'model/')
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 4, 4), dim=16, num_classes=C, extents=tuple(int(e) for e in g['extents']),
                                          depth=2, dim_head=8, mlp_dim=24, heads=int(g['heads']))
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    trace = []
    with wmz['config'].compute_dtype(torch.float32):
        frames, zf = sample.sample_frames(m, g['z0'].cuda(), C, num_frames=g['tokens'].shape[0], sample_topk=int(g['topk']),
                                          uniforms=(g['u_multi'], g['u_mask']), use_graph=use_graph, trace=trace)
    fed = torch.stack(trace).cpu()
    first_bad = (fed != g['fed']).flatten(1).any(1).nonzero()
    assert torch.equal(fed, g['fed']), f'first differing iteration: {int(first_bad[0]) if len(first_bad) else None}'
    assert torch.equal(torch.stack(frames).cpu(), g['tokens'])
    assert torch.equal(zf[:, 0].cpu(), g['tokens'][0]) and torch.equal(zf[:, 1].cpu(), g['tokens'][1])
    # decode like evaluate_model does (:113): the reference's decoder ran in train mode on ae state 'ae0' -> encode of the
    # context mutated the encoder's BN statistics only; the decoder's BN statistics are batch statistics either way
    ae = VqAutoEncoder(embedding_dim=8, num_embeddings=C, downscale_steps=2, hidden_planes=8, in_channels=1)
    ae.load_state_dict(sub(g, 'ae0/'), strict=True)
    ae = ae.cuda().train()
    with wmz['config'].compute_dtype(torch.float32), torch.no_grad():
        img = torch.cat([ae.decode(f) for f in frames], 0)
    assert rel(img, g['images'][2:]) < 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_graphed_training_step_matches_eager(wmz, dtype):
    """DenoiserTrainer.enable_graph: the whole step as one hipGraph.  With r = 0 the corruption is the identity (no mask,
    no redraw), so graphed and eager trainers started from the same weights must walk the same trajectory (lr schedule and
    AdamW bias corrections come from device memory in the graph); then with random r the graph must draw a fresh
    corruption on every replay."""
    torch.manual_seed(8)
    C = 64
    def make():
        torch.manual_seed(9)
        m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=256, num_classes=C, extents=(1, 1, 1), depth=2,
                                              dim_head=128, mlp_dim=256, heads=1).cuda()
        return m
    z = torch.randint(0, C, (2, 3, 16, 16), device='cuda')
    r0 = torch.zeros(2)
    with wmz['config'].compute_dtype(dtype):
        me, mg = make(), make()
        te = wmz['train'].DenoiserTrainer(me, C, lr=1e-3, warmup=2, max_steps=100, distributed=False)
        tg = wmz['train'].DenoiserTrainer(mg, C, lr=1e-3, warmup=2, max_steps=100, distributed=False)
        # enable_graph's warm-up steps run the real step body, then the weights / moments / step count they moved are put
        # back: the graphed trainer starts where the eager one does (r = 0 via a stub sampler)
        class Zero:
            def sample(self, n, generator=None): return torch.zeros(n)
            def update_with_losses(self, *a): pass
        te.sampler, tg.sampler = Zero(), Zero()
        p_before = tg.arena.flat_param.clone()
        tg.enable_graph(z, warmup=2)
        assert torch.equal(tg.arena.flat_param, p_before) and tg.step_count == 0
        assert float(tg.m.abs().sum()) == 0 and float(tg.v.abs().sum()) == 0
        for it in range(5):
            le, ge = te.train_step(z, r=r0)
            lg, gg = tg.train_step(z, r=r0)
            assert abs(le - lg) < (2e-5 if dtype == torch.float32 else 2e-2) * max(1.0, abs(le)), (it, le, lg)
            assert abs(ge - gg) < (1e-3 if dtype == torch.float32 else 5e-2) * max(1.0, abs(ge)), (it, ge, gg)
        assert te.step_count == tg.step_count == 5
        tol = dict(rtol=1e-4, atol=1e-6) if dtype == torch.float32 else dict(rtol=0, atol=3e-3)
        for (n, a), b in zip(me.named_parameters(), mg.parameters()):
            assert torch.allclose(a, b, **tol), n
        # eager inference after graphed steps sees the updated weights (operand caches invalidated)
        with torch.no_grad():
            assert torch.allclose(me(z), mg(z), rtol=1e-3, atol=(1e-4 if dtype == torch.float32 else 5e-2))
        # fresh corruption per replay: the device counter advances, so two replays with the same r differ in loss
        r1 = torch.full((2,), 0.7)
        l1, _ = tg.train_step(z, r=r1)
        c1 = int(tg._g_ctr)
        l2, _ = tg.train_step(z, r=r1)
        assert int(tg._g_ctr) == c1 + 1 and l1 != l2
        assert c1 >= (1 << 39)                            # replays count in their own range of Philox stream ids
        # a library workspace replaced under the graph (a larger eager call grew it): the graph HOLDS the one it was captured
        # with and keeps replaying on it -- no re-capture behind the caller's back (rank-local under data parallelism), and the
        # corruption's stream counter goes on counting
        from world_modelz_amd import ops
        g_before = tg._graph
        dev = z.device
        old = ops._wgrad_ws.get(dev)
        if old is not None:
            ops._wgrad_ws[dev] = torch.empty(old.numel() + 1024, dtype=torch.float32, device=dev)
            c3 = int(tg._g_ctr)
            l3, _ = tg.train_step(z, r=r0)
            assert tg._graph is g_before and l3 == l3 and int(tg._g_ctr) == c3 + 1
            assert any(w is old for w in tg._g_ws)
        # ... and with EVERY library workspace replaced (the counting-sort scratch of the embedding backward and of the VQ statistics
        # zero their counters once and re-zero them per call -- state a replay must find where it left it): the graph replays on
        # the allocations it holds, eager calls move to the new ones, and both produce the step the other does
        for wsd, dtype_ in ((ops._wgrad_ws, torch.float32), (ops._embed_ws, torch.int32), (ops._vq_ws, torch.int32)):
            cur = wsd.get(dev)
            if cur is not None:
                wsd[dev] = (torch.empty if dtype_ == torch.float32 else torch.zeros)(2 * cur.numel() + 4096, dtype=dtype_, device=dev)
        me.load_state_dict(mg.state_dict())
        te.m.copy_(tg.m); te.v.copy_(tg.v); te.step_count = tg.step_count
        from world_modelz_amd import _cast
        _cast.invalidate()
        for it in range(2):
            le, ge = te.train_step(z, r=r0)                 # eager: the new workspaces
            lg, gg = tg.train_step(z, r=r0)                 # replay: the captured ones
            assert tg._graph is g_before
            assert abs(le - lg) < (2e-5 if dtype == torch.float32 else 2e-2) * max(1.0, abs(le)), (it, le, lg)
            assert abs(ge - gg) < (1e-3 if dtype == torch.float32 else 5e-2) * max(1.0, abs(ge)), (it, ge, gg)
        for (n, a), b in zip(me.named_parameters(), mg.parameters()):
            assert torch.allclose(a, b, **tol), n
        # capturing again (a caller's second enable_graph) does not restart the counter either
        c4 = int(tg._g_ctr)
        tg.enable_graph(z, warmup=1)
        assert int(tg._g_ctr) > c4


def test_training_step_full_size_properties(wmz):
    """BASELINE configs[2] / [3] sizes (cfg-3: B = 16 clips of 16x16x16; cfg-4 per GPU: B = 8 of 32x16x16; codebook 1024,
    default denoiser), where the CPU oracle takes minutes per clip: size-independent properties of one bf16 training step.
      * fused training forward == op-by-op forward (loss and every gradient, bf16 tolerance) at full size;
      * gradients are additive over clips: grad(mean loss over B clips) == mean over halves of grad(mean loss over each
        half) -- the exactness of the data-parallel sharding, checked on the real kernels;
      * a clip's per-sample loss does not depend on its batch neighbours (bit-exact)."""
    from world_modelz_amd.train import cross_entropy_rows
    cfg = wmz['config']
    for (B, S) in ((16, 16), (8, 32)):
        torch.manual_seed(77)
        m = wmz['main'].VqVideoDiffusionModel(data_shape=(S, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4,
                                              dim_head=128, mlp_dim=256, heads=1).cuda().train()
        z = torch.randint(0, 1025, (B, S, 16, 16), device='cuda')
        tgt = torch.randint(0, 1024, (B, 16, 16), device='cuda')

        def grads(zz, tt, fused_on):
            cfg.set_fused_training(fused_on)
            m.zero_grad(set_to_none=True)
            with cfg.compute_dtype(torch.bfloat16):
                y = m(zz)
                loss = cross_entropy_rows(y.reshape(-1, 1024), tt.reshape(-1))
                per = loss.view(zz.shape[0], -1).mean(1)
                loss.mean().backward()
            return per.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        try:
            per_f, g_f = grads(z, tgt, True)
            per_u, g_u = grads(z, tgt, False)
            h = B // 2
            per_a, g_a = grads(z[:h], tgt[:h], True)
            per_b, g_b = grads(z[h:], tgt[h:], True)
        finally:
            cfg.set_fused_training(True)
        assert torch.isfinite(per_f).all()
        assert rel(per_f, per_u) < 5e-3
        worst_fu = max(rel(g_f[n], g_u[n]) for n in g_f)
        worst_add = max(rel(0.5 * (g_a[n] + g_b[n]), g_f[n]) for n in g_f)
        print(f'[full-size step B={B} S={S}] fused vs op-by-op worst grad rel {worst_fu:.2e}; clip additivity {worst_add:.2e}')
        assert worst_fu < 1.5e-2                     # measured 5-6e-3
        assert worst_add < 2e-3                      # fp32 accumulation order only (the activations are identical)
        assert torch.equal(per_a, per_f[:h]) and torch.equal(per_b, per_f[h:])
        del m


def test_fused_cross_entropy_vs_torch(wmz):
    tr = wmz['train']
    torch.manual_seed(5)
    for R, C in [(2048, 1024), (77, 50), (33, 8192), (301, 2000), (5, 4096), (9, 9000)]:
        logits = (torch.randn(R, C, device='cuda') * 3).requires_grad_(True)
        target = torch.randint(0, C, (R,), device='cuda')
        ref = torch.nn.functional.cross_entropy(logits, target, reduction='none')
        w = torch.rand(R, device='cuda')
        (gref,) = torch.autograd.grad((ref * w).sum(), logits)
        loss = tr.cross_entropy_rows(logits, target)
        assert torch.allclose(loss, ref, rtol=1e-5, atol=1e-5)
        (g,) = torch.autograd.grad((loss * w).sum(), logits)
        assert torch.allclose(g, gref, rtol=1e-4, atol=1e-7)
        # the kernel can also emit the gradient in the GEMM operand dtype
        from world_modelz_amd import _lib as L
        lse = torch.logsumexp(logits.detach(), -1)
        g16 = torch.empty(R, C, dtype=torch.bfloat16, device='cuda')
        L.call('wmz_ce_bwd', L.ptr(logits.detach()), C, L.ptr(target), L.ptr(lse), L.ptr(w), L.ptr(g16), R, C, L.WMZ_BF16,
               L.stream())
        assert torch.allclose(g16.float(), gref, rtol=2e-2, atol=1e-4)
        # the one-pass form the training step calls (row in registers; C > 8192 falls back to the two launches)
        for dt_code, dt_t, rt in ((L.WMZ_BF16, torch.bfloat16, 2e-2), (L.WMZ_F32, torch.float32, 1e-4)):
            lo, ls = torch.empty(R, device='cuda'), torch.empty(R, device='cuda')
            gg = torch.empty(R, C, dtype=dt_t, device='cuda')
            L.call('wmz_ce_fwd_bwd', L.ptr(logits.detach()), C, L.ptr(target), L.ptr(lo), L.ptr(ls), L.ptr(w), L.ptr(gg), R, C,
                   dt_code, L.stream())
            assert torch.allclose(lo, ref.detach(), rtol=1e-5, atol=1e-5) and torch.allclose(ls, lse, rtol=1e-5, atol=1e-5)
            assert torch.allclose(gg.float(), gref, rtol=rt, atol=1e-4 if dt_t == torch.bfloat16 else 1e-7)


@pytest.mark.parametrize('shape,depth', [((2, 4, 16, 16), 3), ((1, 3, 8, 8), 2), ((2, 2, 5, 5), 2)])
def test_fused_training_forward_matches_op_by_op(shape, depth):
    """bf16 training takes the fused kernels for the forward (wmz_*_train) and the op-by-op backward on what they saved:
    logits and EVERY parameter gradient against the op-by-op forward + backward (same bf16 operands; the differences are
    the fp32-resident residual stream, the folded LayerNorm affines and the recomputed feed-forward pre-activation), and
    against the fp32 oracle.  Shapes: tiled stream (16x16, 8x8) and the row-major fallback (5x5, ragged tiles)."""
    import world_modelz_amd
    from world_modelz_amd import config
    from world_modelz_amd.main import VqVideoDiffusionModel
    from oracle import denoiser as oden
    torch.manual_seed(17)
    B, S, H, W = shape
    C = 96
    m = VqVideoDiffusionModel(data_shape=(S, H, W), dim=256, num_classes=C, extents=(1, 2, 2), depth=depth, dim_head=128,
                              mlp_dim=256, heads=1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'norm' in n or n.endswith('bias'):
                p.add_(0.2 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    z = torch.randint(0, C + 1, (B, S, H, W))
    wgt = torch.randn(B, H, W, C)

    def run(fused_on):
        config.set_fused_training(fused_on)
        m.zero_grad(set_to_none=True)
        with config.compute_dtype(torch.bfloat16):
            y = m(z.cuda())
            (y * wgt.cuda()).sum().backward()
        return y.detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    try:
        y_f, g_f = run(True)
        y_o, g_o = run(False)
    finally:
        config.set_fused_training(True)
    # fp32 oracle gradients by torch autograd on the CPU restatement
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    y_r = oden.denoiser_forward(sdr, z, (1, 2, 2), 1)
    (y_r * wgt).sum().backward()
    assert rel(y_f, y_r.detach()) < 3e-2 and rel(y_f, y_o) < 2e-2
    worst = 0.0
    for n, g in g_f.items():
        ref = sdr[n].grad
        e_f, e_o = rel(g, ref), rel(g_o[n], ref)
        worst = max(worst, e_f)
        assert e_f < 6e-2, (n, e_f, e_o)
        assert e_f < 2.5 * e_o + 2e-2, (n, e_f, e_o)       # no worse than the op-by-op bf16 path, up to noise
    print(f'fused-training gradients: worst relative error vs the fp32 oracle {worst:.3e}')


def test_bulk_operand_refresh_is_exact():
    """wmz_operands_refresh (one launch for every cast / transpose / concatenation the step needs) against the tensor-op
    formulation it replaces: bit-exact, and the refreshed entries are what _cast.operand() then returns."""
    import world_modelz_amd
    from world_modelz_amd import _cast
    torch.manual_seed(5)
    bf = torch.bfloat16
    wk = torch.nn.Parameter(torch.randn(128, 256, device='cuda'))
    wv = torch.nn.Parameter(torch.randn(128, 256, device='cuda'))
    bv = torch.nn.Parameter(torch.randn(128, device='cuda'))
    w1 = torch.nn.Parameter(torch.randn(256, 72, device='cuda'))
    bulk = _cast.BulkOperands()
    d_w1 = bulk.add((w1,), bf, 'w')
    d_w1t = bulk.add((w1,), bf, 'w1T', transpose=True)
    d_kv = bulk.add((wk, wv), bf, 'kv')
    d_kvt = bulk.add((wk, wv), bf, 'kvT', transpose=True)
    d_bkv = bulk.add((bv,), torch.float32, 'bkv', zero_first=True)
    bulk.refresh()
    torch.cuda.synchronize()
    assert torch.equal(d_w1, w1.detach().to(bf))
    assert torch.equal(d_w1t, w1.detach().t().to(bf).contiguous())
    assert torch.equal(d_kv, torch.cat([wk, wv]).detach().to(bf))
    assert torch.equal(d_kvt, torch.cat([wk, wv]).detach().t().to(bf).contiguous())
    assert torch.equal(d_bkv, torch.cat([torch.zeros_like(bv), bv]).detach())
    assert _cast.operand(w1, bf).data_ptr() == d_w1.data_ptr()
    assert _cast.operand((wk, wv), bf, 'kvT', lambda a, b: torch.cat([a, b], dim=0).t()).data_ptr() == d_kvt.data_ptr()
    with torch.no_grad():
        w1.mul_(2.0)                                     # an in-place change torch sees: the stale entry must not be served
    assert torch.equal(_cast.operand(w1, bf), w1.detach().to(bf))


def test_optimizer_state_and_ema_round_trip_through_the_reference_layout(wmz, tmp_path):
    """Checkpoint state of the trainers (main.py:302-309): `optimizer_state_dict()` is torch.optim.AdamW's layout
    -- loaded into a REAL torch AdamW over copies of the weights it continues the trainer's trajectory (same next step from the
    same gradients); `load_optimizer_state_dict` resumes a second trainer identically; `enable_ema` follows ModelEmaV2's law
    (model_ema_v2.py:33-41), eager and graphed; the file written by save_denoiser_checkpoint reads back."""
    from world_modelz_amd import checkpoint
    C = 64
    def make():
        torch.manual_seed(77)
        return wmz['main'].VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=256, num_classes=C, extents=(1, 1, 1), depth=2,
                                                 dim_head=128, mlp_dim=256, heads=1).cuda()
    torch.manual_seed(78)
    z = torch.randint(0, C, (2, 3, 16, 16), device='cuda')
    r0 = torch.zeros(2)
    with wmz['config'].compute_dtype(torch.float32):
        ma = make()
        ta = wmz['train'].DenoiserTrainer(ma, C, lr=1e-3, warmup=0, max_steps=100, distributed=False).enable_ema(0.9)
        w0 = ta.arena.flat_param.clone()
        ema_ref = w0.clone()
        osd0 = ta.optimizer_state_dict()                   # before the first step: torch's AdamW has an empty state then
        assert osd0['state'] == {} and osd0['param_groups'][0]['initial_lr'] == 1e-3
        torch.optim.AdamW([p.detach().clone().requires_grad_(True) for p in ma.parameters()], lr=1e-3).load_state_dict(osd0)
        for _ in range(3):
            ta.train_step(z, r=r0)
            ema_ref = 0.9 * ema_ref + 0.1 * ta.arena.flat_param
        assert torch.allclose(ta.ema_flat, ema_ref, rtol=1e-5, atol=1e-7)
        osd = ta.optimizer_state_dict()
        assert set(osd) == {'state', 'param_groups'} and len(osd['state']) == len(list(ma.parameters()))
        assert float(osd['state'][0]['step']) == 3.0 and osd['param_groups'][0]['betas'] == (0.9, 0.999)
        # (a) a real torch AdamW continues from it
        ref_params = [p.detach().clone().requires_grad_(True) for p in ma.parameters()]
        opt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7)
        opt.load_state_dict({'state': {i: {k: (v.cuda() if k != 'step' else v) for k, v in st.items()} for i, st in osd['state'].items()},
                             'param_groups': osd['param_groups']})
        # (b) a second trainer resumes from it
        mb = make()
        mb.load_state_dict(ma.state_dict())
        tb = wmz['train'].DenoiserTrainer(mb, C, lr=1e-3, warmup=0, max_steps=100, distributed=False)
        tb.load_optimizer_state_dict(osd)
        assert tb.step_count == 3
        # one more step everywhere, same gradients (r = 0: no corruption randomness)
        ta.arena.zero_grad()
        from world_modelz_amd.train import corrupt_last_frame
        zc, tgt = corrupt_last_frame(z, r0, C)
        ta.forward_backward(zc, tgt)
        grads = [p.grad.detach().clone() for p in ma.parameters()]
        lr4 = wmz['train'].lr_at(4, 1e-3, 0, 100)
        for g_ in opt.param_groups:
            g_['lr'] = lr4
        for p, g in zip(ref_params, grads):
            p.grad = g
        opt.step()
        ta.optimizer_step()
        tb.train_step(z, r=r0)
        for p, q, w in zip(ma.parameters(), ref_params, mb.parameters()):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
            assert torch.allclose(p, w, rtol=1e-4, atol=1e-6)
        # the checkpoint file in the reference's layout
        import argparse
        path = str(tmp_path / 'd_checkpoint_0000004.pth')
        checkpoint.save_denoiser_checkpoint(path, step=4, lr=[lr4], model=ma, opt=argparse.Namespace(dim=256), ema_model=None,
                                            optimizer_state=ta.optimizer_state_dict())
        data = checkpoint.read(path)
        assert data['step'] == 4 and float(data['optimizer_state_dict']['state'][0]['step']) == 4.0
        esd = ta.ema_state_dict()
        assert set(esd) == set(ma.state_dict()) and not torch.equal(esd['logit_proj.weight'], ma.logit_proj.weight.cpu())
    # EMA inside the captured step
    with wmz['config'].compute_dtype(torch.bfloat16):
        mg = make()
        tg = wmz['train'].DenoiserTrainer(mg, C, lr=1e-3, warmup=0, max_steps=100, distributed=False).enable_ema(0.5)
        tg.enable_graph(z)
        e = tg.arena.flat_param.clone()
        assert torch.equal(tg.ema_flat, e)
        for _ in range(2):
            tg.train_step(z, r=r0)
            e = 0.5 * e + 0.5 * tg.arena.flat_param
        assert torch.allclose(tg.ema_flat, e, rtol=1e-5, atol=1e-7)
    # ModelEmaV2 averages EVERY state_dict value: the VQ auto-encoder's buffers (BatchNorm running statistics, codebook, cluster
    # sizes) follow the same law, not the live values
    with wmz['config'].compute_dtype(torch.float32):
        torch.manual_seed(5)
        from world_modelz_amd.train_vqae import VqAutoEncoder
        ae = VqAutoEncoder(embedding_dim=16, num_embeddings=32, downscale_steps=2, hidden_planes=24).cuda()
        vt = wmz['train'].VqaeTrainer(ae, distributed=False).enable_ema(0.75)
        keys = [k for k in ae.state_dict() if k.endswith('running_mean') or k in ('vq.embedding', 'vq.cluster_size')]
        ref = {k: ae.state_dict()[k].clone() for k in keys}
        fr = torch.rand(4, 3, 32, 32, device='cuda')
        for _ in range(2):
            vt.train_step(fr)
            for k in keys:
                ref[k] = 0.75 * ref[k] + 0.25 * ae.state_dict()[k]
        esd = vt.ema_state_dict()
        assert set(esd) == set(ae.state_dict())
        for k in keys:
            assert torch.allclose(esd[k], ref[k].cpu(), rtol=1e-5, atol=1e-7), k
            assert not torch.equal(esd[k], ae.state_dict()[k].cpu()), k


def test_overlapped_allreduce_path_runs_on_rccl_world_of_one():
    """The RCCL world-of-one checks below, in a FRESH child process: a host-side crash inside an RCCL-capturing capture_end()
    (round 4: streams re-used from torch's pool of 32 after ~250 tests' worth of new streams -- fixed by config.shared_stream)
    takes every test behind it down with it; here it would cost one test.  WMZ_RCCL_CHILD=1 runs the body in this process."""
    import os
    import subprocess
    import sys
    if os.environ.get('WMZ_RCCL_CHILD') == '1':
        return _rccl_world_of_one_body()
    r = subprocess.run([sys.executable, '-m', 'pytest', f'{os.path.abspath(__file__)}::test_overlapped_allreduce_path_runs_on_rccl_world_of_one',
                        '-m', 'gpu', '-q', '-s', '-p', 'no:cacheprovider'], env=dict(os.environ, WMZ_RCCL_CHILD='1', HSA_ENABLE_IPC_MODE_LEGACY='0'),
                       capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for line in r.stdout.splitlines():
        if line.startswith('[ddp'):
            print(line)
    if r.returncode != 0:                                          # (keep the whole story: the head of stderr names what aborted)
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        if os.path.isdir(out):
            with open(os.path.join(out, 'rccl_child_failure.log'), 'w') as f:
                f.write(r.stdout + '\n=== stderr ===\n' + r.stderr)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[:3000] + '\n...\n' + r.stderr[-1500:]


def _rccl_world_of_one_body():
    """SURVEY 8(e) on ONE GPU: a world-1 `nccl` (= RCCL) process group, DenoiserTrainer(distributed=True).  The HIP backward
    accumulates straight into the flat gradient arena and tells the reducer (`_wmz_ready`), which all-reduces each per-layer
    bucket on a SIDE stream as soon as its last gradient has landed.  Checked: every bucket's collective is enqueued by the
    backward itself (not by finish()), deepest layer first, the head's bucket before the layers', the embeddings' last; and
    the step equals the non-distributed trainer's step."""
    import os
    import torch.distributed as dist
    from world_modelz_amd import config, main, train
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        def make():
            torch.manual_seed(21)
            return main.VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=256, num_classes=64, extents=(1, 1, 1), depth=3,
                                              dim_head=128, mlp_dim=256, heads=1).cuda()
        z = torch.randint(0, 64, (2, 3, 16, 16), device='cuda')
        r = torch.zeros(2)
        with config.compute_dtype(torch.bfloat16):
            md, ms = make(), make()
            td = train.DenoiserTrainer(md, 64, lr=1e-3, warmup=0, distributed=True)
            ts = train.DenoiserTrainer(ms, 64, lr=1e-3, warmup=0, distributed=False)
            red = td.reducer
            assert red is not None and red.active and red.world == 1
            assert len(red.buckets) == 3 + 2                      # one per layer + embeddings + head
            launched_by_backward = []
            orig_finish = red.finish

            def finish():
                launched_by_backward.append(list(red.order))      # what the hooks enqueued before finish() ran
                return orig_finish()
            red.finish = finish
            red.enable_timing()
            for _ in range(2):
                ld, gd = td.train_step(z, r=r)
                ls, gs = ts.train_step(z, r=r)
                assert abs(ld - ls) < 1e-3 * max(1.0, abs(ls)) and abs(gd - gs) < 2e-2 * max(1.0, gs)
        for order in launched_by_backward:
            assert sorted(order) == list(range(5)), f'buckets left for finish(): {order}'
            assert order[0] == 4 and order[-1] == 0               # head first, embeddings last
            assert order[1:4] == [3, 2, 1]                        # layers back to front
        for (n, a), b in zip(md.named_parameters(), ms.parameters()):
            assert torch.allclose(a, b, rtol=0, atol=3e-3), n
        torch.cuda.synchronize()
        tm = red.timing_summary()                                 # what bench.py reports as grad_allreduce_overlap
        assert tm['collectives'] == 2 * 5 and tm['allreduce_ms'] > 0 and 0.0 <= tm['overlap_fraction'] <= 1.0, tm
        red.enable_timing(False)
        # ---- the same data-parallel step as ONE hipGraph: the RCCL all-reduces are captured on the reducer's side stream
        # (fork at the bucket's last gradient, join in finish()); trajectory == the eager data-parallel trainer's
        import time
        with config.compute_dtype(torch.bfloat16):
            me, mg = make(), make()
            te = train.DenoiserTrainer(me, 64, lr=1e-3, warmup=0, distributed=True)
            tg = train.DenoiserTrainer(mg, 64, lr=1e-3, warmup=0, distributed=True)
            assert tg.reducer is not None and tg.reducer.active
            tg.enable_graph(z)
            assert tg._graph is not None and tg.step_count == 0
            for it in range(3):
                le, ge = te.train_step(z, r=r)
                lg, gg = tg.train_step(z, r=r)
                assert abs(le - lg) < 2e-2 * max(1.0, abs(le)), (it, le, lg)
                assert abs(ge - gg) < 5e-2 * max(1.0, abs(ge)), (it, ge, gg)
            for (n, a), b in zip(me.named_parameters(), mg.parameters()):
                assert torch.allclose(a, b, rtol=0, atol=3e-3), n
            # every bucket's collective sits inside the captured step (the hooks launched all five during the capture)
            assert sorted(tg.reducer.last_order) == list(range(5)) and tg.reducer.last_order[0] == 4
            wall = {}
            for name, t in (('eager', te), ('graph', tg)):
                for _ in range(5):
                    t.train_step(z, r=r)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    t.train_step(z, r=r)
                torch.cuda.synchronize()
                wall[name] = (time.perf_counter() - t0) / 30 * 1e3
        print(f'[ddp world-of-one, reducer on] eager {wall["eager"]:.3f} ms/step, hipGraph {wall["graph"]:.3f} ms/step')
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        if os.path.isdir(out_dir):
            import json
            with open(os.path.join(out_dir, 'ddp_graph_world1.json'), 'w') as f:
                json.dump({'shape': 'B=2 x 3x16x16, depth 3, default widths, bf16, RCCL world of one, 5 buckets',
                           'eager_ms_per_step': wall['eager'], 'graph_ms_per_step': wall['graph']}, f)
        assert wall['graph'] <= wall['eager'] * 1.05
        # ---- the other two training paths under the data-parallel capture: dim 160 = the op-by-op path (a width without fused
        # per-token kernels): its arena-bound weight gradients leave as batches on a side branch (ops.linear_wgrad), and the
        # reducer must wait for THAT stream too before it all-reduces a bucket; dim 96 = the chain kernels of the published widths
        # (round 4: forward and backward on csrc/layer_chain*.hip, gradients announced per layer like the default widths').
        # graphed data-parallel step == eager data-parallel step == single-process step
        for dim_, batches_expected in ((160, True), (96, False)):
            def make_w():
                torch.manual_seed(23)
                return main.VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=dim_, num_classes=64, extents=(1, 1, 1), depth=3,
                                                  dim_head=128, mlp_dim=256, heads=1).cuda()
            with config.compute_dtype(torch.bfloat16):
                me, mg, ms = make_w(), make_w(), make_w()
                te = train.DenoiserTrainer(me, 64, lr=1e-3, warmup=0, distributed=True)
                tg = train.DenoiserTrainer(mg, 64, lr=1e-3, warmup=0, distributed=True)
                ts = train.DenoiserTrainer(ms, 64, lr=1e-3, warmup=0, distributed=False)
                assert (tg.chain_packs is not None) == (not batches_expected)
                from world_modelz_amd import ops as _ops
                issued = []
                orig_issue = _ops._issue_pending

                def issue():
                    issued.append(len(_ops._pending))
                    return orig_issue()
                _ops._issue_pending = issue
                try:
                    tg.enable_graph(z)
                finally:
                    _ops._issue_pending = orig_issue
                if batches_expected:
                    assert sum(issued) >= 3 * 4, issued               # the capture did queue the layers' weight gradients as batches
                for it in range(3):
                    le, ge = te.train_step(z, r=r)
                    lg, gg = tg.train_step(z, r=r)
                    ls, gs = ts.train_step(z, r=r)
                    assert abs(le - lg) < 2e-2 * max(1.0, abs(le)) and abs(ls - lg) < 2e-2 * max(1.0, abs(ls)), (dim_, it, le, lg, ls)
                    assert abs(ge - gg) < 5e-2 * max(1.0, abs(ge)) and abs(gs - gg) < 5e-2 * max(1.0, abs(gs)), (dim_, it, ge, gg, gs)
                assert sorted(te.reducer.last_order) == list(range(len(te.reducer.buckets))) and te.reducer.last_order[0] == len(te.reducer.buckets) - 1
                for (n, a), b, c in zip(me.named_parameters(), mg.parameters(), ms.parameters()):
                    assert torch.allclose(a, b, rtol=0, atol=3e-3) and torch.allclose(c, b, rtol=0, atol=3e-3), (dim_, n)
                for t_ in (tg, te, ts):
                    if getattr(t_, '_graph', None) is not None:
                        t_._graph = None
                torch.cuda.synchronize()
        # ---- the VQ-AE trainer under the data-parallel capture (round 5: VqaeTrainer.enable_graph accepts a reducer): the gradient
        # buckets' all-reduces AND the VQ EMA statistics' (counts, dw inside VectorQuantizerEMA.forward: SURVEY 8e) are graph nodes;
        # graphed data-parallel step == eager data-parallel step == single-process step
        from world_modelz_amd.train_vqae import VqAutoEncoder

        def make_ae():
            torch.manual_seed(29)
            return VqAutoEncoder(embedding_dim=16, num_embeddings=64, downscale_steps=2, hidden_planes=24).cuda()
        frames = torch.rand(8, 3, 32, 32, device='cuda')
        with config.compute_dtype(torch.bfloat16):
            ae_e, ae_g, ae_s = make_ae(), make_ae(), make_ae()
            ve = train.VqaeTrainer(ae_e, distributed=True)
            vg = train.VqaeTrainer(ae_g, distributed=True)
            vs = train.VqaeTrainer(ae_s, distributed=False)
            assert vg.reducer is not None and vg.reducer.active and ae_g.vq.sync_stats
            vg.enable_graph(frames, warmup=0)
            assert vg._graph is not None
            for it in range(3):
                oe, og, os_ = ve.train_step(frames), vg.train_step(frames), vs.train_step(frames)
                for j, (a, b, c) in enumerate(zip(oe, og, os_)):          # loss, reconstruction loss, latent loss, perplexity
                    tol = 5e-2 if j == 3 else 2e-2                       # (the perplexity counts nearest-code flips of bf16 latents)
                    assert abs(a - b) < tol * max(1.0, abs(a)) and abs(c - b) < tol * max(1.0, abs(c)), (it, oe, og, os_)
            assert sorted(vg.reducer.last_order) == list(range(len(vg.reducer.buckets)))
            for (n, a), b, c in zip(ae_e.named_parameters(), ae_g.parameters(), ae_s.parameters()):
                assert torch.allclose(a, b, rtol=0, atol=3e-3) and torch.allclose(c, b, rtol=0, atol=3e-3), n
            # (the EMA codebook follows the encoder's bf16 latents: a handful of nearest-code flips between launch orders move rows by ~1e-3)
            #  -- and a rarely used code's row by much more: (batch sum) / (EMA count): compared on average, not row by row)
            assert float((ae_e.vq.embedding - ae_g.vq.embedding).abs().median()) < 1e-3 and float((ae_s.vq.embedding - ae_g.vq.embedding).abs().median()) < 1e-3
            print('[ddp world-of-one] graphed data-parallel VQ-AE step == eager == single-process')
            for t_ in (vg, ve, vs):
                if getattr(t_, '_graph', None) is not None:
                    t_._graph = None
            torch.cuda.synchronize()
    finally:
        # (captured graphs hold RCCL kernels: let go of them and drain the device before the communicator is torn down -- a
        #  failing assertion above must surface as that assertion, not as an abort inside destroy_process_group)
        import gc
        for name in ('tg', 'te', 'td', 'ts', 'vg', 've', 'vs'):
            t = locals().get(name)
            if t is not None and getattr(t, '_graph', None) is not None:
                t._graph = None
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()


@pytest.mark.parametrize('dim,mlp', [(96, 256), (384, 512)])
def test_published_widths_train_on_the_chain_kernel_vs_oracle(wmz, dim, mlp):
    """The reference's published widths (results/README.md:7-22) TRAIN on csrc/layer_chain.hip's training forward (round 4): loss
    and every parameter gradient of one step against the fp32 oracle's autograd (bf16 bound as for the default widths) and against
    the op-by-op path (the two paths' rounding differences); the weight streams are one gather of the flat arena, rebuilt per step."""
    from oracle import train_step as ots
    torch.manual_seed(42)
    C = 64
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(4, 16, 16), dim=dim, num_classes=C, extents=(3, 1, 1), depth=2, dim_head=128,
                                          mlp_dim=mlp, heads=1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    z = torch.randint(0, C + 1, (2, 4, 16, 16))
    target = torch.randint(0, C, (2, 16, 16))
    _, _, loss_ref, grads_ref = ots.step_grads(sd, z, target, (3, 1, 1), 1)
    m = m.cuda()
    cfg = wmz['config']
    from conftest import chain_policy
    with cfg.compute_dtype(torch.bfloat16), chain_policy('always'):
        tr = wmz['train'].DenoiserTrainer(m, C, lr=1e-3, warmup=0, max_steps=100, distributed=False)
        assert tr.chain_packs is not None
        got = {}
        # 'chain': forward and backward on the chain kernels; 'chain_fwd': chain forward, op-by-op backward; 'ops': op by op
        for mode, (fused_on, fused_bwd) in (('chain', (True, True)), ('chain_fwd', (True, False)), ('ops', (False, True))):
            cfg.set_fused_training(fused_on)
            cfg.set_fused_backward(fused_bwd)
            try:
                tr.arena.zero_grad()
                _, mean = tr.forward_backward(z.cuda(), target.cuda())
            finally:
                cfg.set_fused_training(True)
                cfg.set_fused_backward(True)
            got[mode] = (float(mean), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
        loss_c, g_c = got['chain']
        loss_o, g_o = got['ops']
        assert abs(loss_c - float(loss_ref)) < 2e-2 and abs(loss_c - loss_o) < 2e-2
        for mode in ('chain', 'chain_fwd'):
            g_m = got[mode][1]
            worst = max((float((g_m[n] - grads_ref[n].cuda()).norm() / (grads_ref[n].norm() + 1e-12)), n) for n in g_m)
            worst_o = max((float((g_m[n] - g_o[n]).norm() / (g_o[n].norm() + 1e-12)), n) for n in g_m)
            print(f'dim {dim} [{mode}]: gradients vs oracle {worst[0]:.3e} ({worst[1]}), vs op-by-op {worst_o[0]:.3e} ({worst_o[1]})')
            assert worst[0] < 6e-2, (mode, worst)
            assert worst_o[0] < 6e-2, (mode, worst_o)
        # the packed streams follow the weights: one optimizer step, then the same forward must see the new weights
        tr.optimizer_step()
        tr.arena.zero_grad()
        _, mean2 = tr.forward_backward(z.cuda(), target.cuda())
        cfg.set_fused_training(False)
        try:
            tr.arena.zero_grad()
            _, mean3 = tr.forward_backward(z.cuda(), target.cuda())
        finally:
            cfg.set_fused_training(True)
        assert float(mean2) != loss_c and abs(float(mean2) - float(mean3)) < 2e-2


def _two_rank_training(tmp_path, backend, port, dtype='bfloat16', atol=3e-3):
    """Two data-parallel ranks as fresh child processes (nothing has touched the GPU in them before), `backend` 'nccl' (= RCCL,
    one card per rank) or 'gloo' (both ranks on cuda:0: the in-place gradient writes, the `_wmz_ready` notifications, the
    bucket order and the VQ statistics all-reduce run with the real HIP kernels at world 2 on a ONE-GPU box).  Checked in the
    ranks: replicas bit-identical after 3 steps, both codebooks identical; here: equal to the single-process run on the union
    batch (gradient = mean over the two shards; VQ statistics = those of the whole batch)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'rank.py'
    script.write_text(f'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
rank = int(os.environ['RANK'])
dev = rank if {backend!r} == 'nccl' else 0
torch.cuda.set_device(dev)
if {backend!r} == 'nccl':
    dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
else:
    dist.init_process_group('gloo')
from world_modelz_amd import config, main, train
from world_modelz_amd.vq import VectorQuantizerEMA
torch.manual_seed(5)
m = main.VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=256, num_classes=64, extents=(1, 1, 1), depth=2, dim_head=128, mlp_dim=256, heads=1).cuda()
torch.manual_seed(6)
z = torch.randint(0, 64, (4, 3, 16, 16), device='cuda')
with config.compute_dtype(torch.{dtype}):
    tr = train.DenoiserTrainer(m, 64, lr=1e-3, warmup=0, distributed=True)
    assert tr.reducer.world == 2 and len(tr.reducer.buckets) == 4
    for _ in range(3):
        tr.train_step(z[2 * rank:2 * rank + 2], r=torch.zeros(2))
        assert sorted(tr.reducer.last_order) == [0, 1, 2, 3] and tr.reducer.last_order[0] == 3, tr.reducer.last_order
flat = tr.arena.flat_param.clone()
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
assert torch.equal(other[0], other[1]), 'replicas diverged'
# VQ EMA statistics (vq.py:42-65) all-reduced before the update
torch.manual_seed(12)
q = VectorQuantizerEMA(16, 32).cuda()
q.sync_stats = True
q.train()
g = torch.Generator().manual_seed(13)
for _ in range(3):
    x = torch.randn(2, 8, 8, 16, generator=g).cuda()
    q(x[rank:rank + 1])
cb = q.embedding.clone()
both = [torch.empty_like(cb) for _ in range(2)]
dist.all_gather(both, cb)
assert torch.equal(both[0], both[1]), 'codebooks diverged'
if rank == 0:
    torch.save({{'flat': flat.cpu(), 'embedding': cb.cpu(), 'cluster_size': q.cluster_size.cpu(), 'act': q.activation_count.cpu()}},
               {str(tmp_path / "dp.pt")!r})
dist.destroy_process_group()
''')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    from world_modelz_amd import config, main, train
    from world_modelz_amd.vq import VectorQuantizerEMA
    torch.manual_seed(5)
    m = main.VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=256, num_classes=64, extents=(1, 1, 1), depth=2, dim_head=128,
                                   mlp_dim=256, heads=1).cuda()
    torch.manual_seed(6)
    z = torch.randint(0, 64, (4, 3, 16, 16), device='cuda')
    with config.compute_dtype(getattr(torch, dtype)):
        tr = train.DenoiserTrainer(m, 64, lr=1e-3, warmup=0, distributed=False)
        for _ in range(3):
            tr.train_step(z, r=torch.zeros(4))
    dp = torch.load(str(tmp_path / 'dp.pt'))
    err = float((tr.arena.flat_param.cpu() - dp['flat']).abs().max())
    print(f'[two ranks, {backend}, {dtype}] max |w(union batch) - w(2 ranks)| after 3 steps: {err:.2e}')
    assert err <= atol, err
    torch.manual_seed(12)
    q = VectorQuantizerEMA(16, 32).cuda()
    q.train()
    g = torch.Generator().manual_seed(13)
    for _ in range(3):
        q(torch.randn(2, 8, 8, 16, generator=g).cuda())
    assert torch.allclose(q.embedding.cpu(), dp['embedding'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(q.cluster_size.cpu(), dp['cluster_size'], rtol=1e-6, atol=1e-7)
    assert torch.equal(q.activation_count.cpu(), dp['act'])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the driver\'s multi-GPU tier)')
def test_two_rank_rccl_training_matches_single_process(tmp_path):
    """Two ranks over RCCL, one card each."""
    _two_rank_training(tmp_path, 'nccl', 29541)


def test_two_rank_training_on_one_card_matches_single_process(tmp_path):
    """The same two-rank run with both ranks on cuda:0 and gloo carrying the collectives: everything of the data-parallel path
    except RCCL itself, on the one-GPU box."""
    _two_rank_training(tmp_path, 'gloo', 29543)


def test_two_rank_training_on_one_card_equals_the_union_batch_in_fp32(tmp_path):
    """The same two-rank run in the fp32 compute mode: without bf16 rounding between them, the weights after three data-parallel
    steps must equal those of ONE process stepping on the union batch to 1e-5 (gradient of the mean loss = mean of the shards'
    gradients: the exactness of the sharding + all-reduce + 1/world in the AdamW pass, not just replica-vs-replica identity)."""
    _two_rank_training(tmp_path, 'gloo', 29545, dtype='float32', atol=1e-5)


@pytest.mark.parametrize('C,dim,mlp,dtype', [(37, 256, 256, torch.bfloat16), (1001, 256, 256, torch.bfloat16), (37, 64, 96, torch.float32),
                                             (1003, 64, 96, torch.bfloat16), (45, 96, 256, torch.bfloat16)])
def test_vocabulary_that_is_no_multiple_of_8(wmz, C, dim, mlp, dtype):
    """The reference takes any --num_embeddings (main.py:404); the GEMM kernels' granule is 8 elements.  The fused linear +
    cross-entropy (default / chain widths) and the op-by-op logits backward run a class count that is no multiple of 8 on
    zero-padded operands (padding classes at probability exactly 0): logits, loss and every gradient of one training step against
    the fp32 oracle's autograd."""
    from conftest import chain_policy
    from oracle import denoiser as oden
    from oracle import train_step as ots
    torch.manual_seed(C + dim)
    heads = 1 if dim != 64 else 2
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=dim, num_classes=C, extents=(1, 1, 1), depth=2,
                                          dim_head=128 if dim != 64 else 32, mlp_dim=mlp, heads=heads)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    z = torch.randint(0, C + 1, (2, 3, 16, 16))
    target = torch.randint(0, C, (2, 16, 16))
    ref_logits = oden.denoiser_forward(sd, z, (1, 1, 1), heads)
    _, _, loss_ref, grads_ref = ots.step_grads(sd, z, target, (1, 1, 1), heads)
    m = m.cuda()
    cfg = wmz['config']
    with cfg.compute_dtype(dtype), chain_policy('always'):
        with torch.no_grad():
            y = m(z.cuda())
        assert y.shape == (2, 16, 16, C)
        tr = wmz['train'].DenoiserTrainer(m, C, lr=1e-3, warmup=0, max_steps=100, distributed=False)
        tr.arena.zero_grad()
        _, mean = tr.forward_backward(z.cuda(), target.cuda())
    e_y = rel(y, ref_logits)
    worst = max((float((p.grad.detach().cpu() - grads_ref[n]).norm() / (grads_ref[n].norm() + 1e-12)), n) for n, p in m.named_parameters())
    print(f'C = {C}, dim {dim} {str(dtype)[6:]}: logits {e_y:.2e}, loss diff {abs(float(mean) - float(loss_ref)):.1e}, worst gradient {worst[0]:.2e} ({worst[1]})')
    f32 = dtype == torch.float32
    assert e_y < (1e-5 if f32 else 1e-2)
    assert abs(float(mean) - float(loss_ref)) < (1e-5 if f32 else 2e-2)
    assert worst[0] < (2e-4 if f32 else 6e-2), worst
