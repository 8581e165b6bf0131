"""Model-level parity of the bf16 conv route -- the one bench.py times (csrc/conv_direct.hip, conv_point.hip, conv_wgrad.hip,
BatchNorm finalised by its consumer; autoencoder.py:18-152, train_vqae.py:33-49, main.py:229-237 with BatchNorm in training
mode) -- against the fp32 CPU oracle: latents, tokens with every disagreement accounted for, running statistics, and the
gradients of one VqAutoEncoder.forward + backward against the oracle's autograd.  Default sizes: E 64 / P 128 / 2 down-scale
steps, 64 x 64 frames.

Token accounting.  The HIP arg-min is bit-exact on equal inputs (tests/test_kernels_gpu.py), so the tokens of a bf16 encoder
differ from the reference's only where the bf16 latent x' = x + delta changes the nearest code.  For such a position the
triangle inequality bounds how far apart the two codes can be seen from the ORACLE's latent x:
        ||x - c_picked|| - ||x - c_best||  <=  2 ||delta||
(the picked code is nearest to x'), which the tests check per position with the measured delta: every disagreement is a
near-tie at the scale of the latent error, none is anything else."""
import pytest
import torch

from conftest import recorded_calls

pytestmark = pytest.mark.gpu

from oracle import autoencoder as oae        # noqa: E402
from oracle import vq as ovq                 # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _default_vqae(seed, C):
    from world_modelz_amd.train_vqae import VqAutoEncoder
    torch.manual_seed(seed)
    return VqAutoEncoder(embedding_dim=64, num_embeddings=C, downscale_steps=2, hidden_planes=128).cuda()


def _cpu_state(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def account_for_token_disagreements(tok, lat_hip, lat_ref, codebook):
    """tok: the HIP route's tokens [N]; lat_hip / lat_ref [N, E] (fp32, CPU); codebook [C, E].  Returns (agreement, n_diff).
    Asserts (i) the HIP tokens ARE the oracle's arg-min of the HIP latents (bit-exact search on equal inputs) and (ii) every
    disagreement with the oracle's tokens obeys the triangle bound of the module docstring."""
    emb = codebook[None]
    tok_ref = ovq.encode(lat_ref, emb).reshape(-1)
    tok_same_in = ovq.encode(lat_hip, emb).reshape(-1)
    assert torch.equal(tok, tok_same_in), 'the codebook search is not bit-exact on its own inputs'
    bad = (tok != tok_ref).nonzero().reshape(-1)
    agree = 1.0 - bad.numel() / tok.numel()
    if bad.numel():
        x, xp = lat_ref[bad].double(), lat_hip[bad].double()
        d_best = (x - codebook[tok_ref[bad]].double()).norm(dim=-1)
        d_pick = (x - codebook[tok[bad]].double()).norm(dim=-1)
        delta = (xp - x).norm(dim=-1)
        slack = d_pick - d_best - 2 * delta
        assert float(slack.max()) <= 1e-6, f'a token differs beyond what its latent error explains (slack {float(slack.max()):.3e})'
        print(f'[tokens] {bad.numel()} of {tok.numel()} differ; max (d_pick - d_best) / d_best = '
              f'{float(((d_pick - d_best) / d_best).max()):.3e}, all within 2 |delta|')
    return agree, int(bad.numel())


@pytest.mark.parametrize('n_frames,C', [(32, 1024), (16, 512)])
def test_bf16_direct_route_frame_encoder_vs_oracle(n_frames, C):
    """main.py:229-237 on the benched route: latents, tokens, BatchNorm running statistics against the fp32 oracle."""
    from world_modelz_amd import config, ops
    m = _default_vqae(41, C)
    m.train()                                                      # quirk Q3: the frozen AE is never .eval()-ed
    sd = _cpu_state(m)
    torch.manual_seed(42)
    frames = torch.rand(n_frames, 3, 64, 64)
    p = {k: v.clone() for k, v in sd.items()}
    lat_ref = oae.encoder_forward(p, frames, training=True).permute(0, 2, 3, 1).contiguous()      # [B,16,16,64]; p's statistics moved
    assert ops.DIRECT_CONV and ops.BN_LAZY
    with config.compute_dtype(torch.bfloat16), torch.no_grad(), recorded_calls() as seen:
        lat = m._latents(frames.cuda())
        tok = m.vq.encode(lat).reshape(-1).cpu()
    # the route under test is the one that ran: direct 3x3 (both strides), streaming small-K with the BatchNorm prologue, nothing
    # on the implicit-GEMM fallback
    assert 'wmz_conv3x3_direct_fwd_strided' in seen and 'wmz_conv_point_fwd_bn' in seen, set(seen)
    assert not any(n.startswith('wmz_conv2d_nhwc_fwd') for n in seen), set(seen)
    assert 'wmz_bn_finalize' not in seen
    lat = lat.float().cpu()
    e_lat = rel(lat, lat_ref)
    print(f'[latents] bf16 direct route vs fp32 oracle: rel {e_lat:.3e} ({n_frames} frames)')
    assert lat.shape == lat_ref.shape and e_lat < 4e-2, e_lat
    agree, n_bad = account_for_token_disagreements(tok, lat.reshape(-1, 64), lat_ref.reshape(-1, 64), sd['vq.embedding'][0])
    print(f'[tokens] agreement with the fp32 oracle: {agree:.4f}')
    assert agree > 0.93, agree
    # running statistics after ONE pass (momentum 0.1 from the initial 0 / 1) against the oracle's
    for k, v in m.state_dict().items():
        if not k.startswith('encoder.'):
            continue
        if k.endswith('running_mean'):
            assert torch.allclose(v.cpu(), p[k], rtol=2e-2, atol=2e-3), (k, float((v.cpu() - p[k]).abs().max()))
        elif k.endswith('running_var'):
            assert torch.allclose(v.cpu(), p[k], rtol=2e-2, atol=1e-4), (k, float((v.cpu() - p[k]).abs().max()))
        elif k.endswith('num_batches_tracked'):
            assert int(v) == int(p[k]) == 1, k


def test_bf16_vqae_forward_backward_vs_oracle_autograd():
    """train_vqae.py:139-150 on the bf16 route (direct forward kernels, direct 3x3 weight gradient, BatchNorm backward kernels):
    recon, losses and EVERY parameter gradient against torch autograd over the fp32 oracle.  A bf16 encoder flips a few code
    assignments at near-ties (accounted for as above); the gradients are compared on the SAME assignment (oracle.vq.forward's
    test knob), and additionally bounded against the oracle's own assignment."""
    from world_modelz_amd import config
    m = _default_vqae(43, 256)
    m.train()
    torch.manual_seed(44)
    frames = torch.rand(16, 3, 64, 64)
    sd = _cpu_state(m)
    # a codebook of actual latents (what a trained VQ-AE has), so that assignments are decided by the data rather than by the
    # near-ties of 256 random directions
    with torch.no_grad():
        lat0 = oae.encoder_forward({k: v.clone() for k, v in sd.items()}, frames, training=True).permute(0, 2, 3, 1).reshape(-1, 64)
        pick = torch.randperm(lat0.shape[0])[:256]
        sd['vq.embedding'] = (lat0[pick] + 0.02 * lat0.std() * torch.randn(256, 64))[None].contiguous()
        m.vq.embedding.copy_(sd['vq.embedding'].cuda())

    def loss_of(recon, x, ll):
        return torch.nn.functional.mse_loss(recon, x) + 0.25 * ll

    with config.compute_dtype(torch.bfloat16), recorded_calls() as seen:
        x = frames.cuda().requires_grad_(True)
        h = m._latents(x)
        q, enc, ll, ppl = m.vq.forward(h)
        recon = m._decode_latents(q)
        loss = loss_of(recon, x, ll)
        loss.backward()
    assert 'wmz_conv3x3_direct_fwd_strided' in seen and 'wmz_conv_point_fwd_bn' in seen, set(seen)
    tok = enc.indices.reshape(-1).cpu()
    agree, n_bad = account_for_token_disagreements(tok, h.detach().float().cpu().reshape(-1, 64), lat0, sd['vq.embedding'][0])
    print(f'[tokens] training forward, agreement with the oracle: {agree:.4f}')
    assert agree > 0.97, agree

    def oracle_run(assign):
        leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k and not k.startswith('vq.')
                      else v.clone()) for k, v in sd.items()}
        xo = frames.clone().requires_grad_(True)
        rec_o, ll_o, ppl_o = oae.vqae_forward(oae.with_vq_stats(leaves), xo, training=True, assign=assign)
        lo = loss_of(rec_o, xo, ll_o)
        lo.backward()
        return leaves, xo, rec_o, ll_o, lo

    results = {}
    for label, assign in (('same assignment', tok.view(-1, 1)), ('oracle assignment', None)):
        leaves, xo, rec_o, ll_o, lo = oracle_run(assign)
        floor = 1e-3 * max(float(leaves['encoder.' + n].grad.norm()) for n, _ in m.encoder.named_parameters())
        errs = []
        for n, prm in m.named_parameters():
            ref = leaves[n].grad
            assert prm.grad is not None and ref is not None, n
            # (a bias in front of a training-mode BatchNorm has an exactly-zero true gradient: measure against the typical
            #  gradient magnitude rather than against rounding noise)
            errs.append((float((prm.grad.float().cpu() - ref).norm() / max(float(ref.norm()), floor)), n))
        worst = max(errs)
        cos = min(float(torch.nn.functional.cosine_similarity(prm.grad.float().cpu().reshape(1, -1), leaves[n].grad.reshape(1, -1)))
                  for n, prm in m.named_parameters() if float(leaves[n].grad.norm()) > floor)
        results[label] = dict(recon=rel(recon, rec_o), loss=abs(float(loss) - float(lo)) / abs(float(lo)),
                              dx=rel(x.grad, xo.grad), worst=worst, median=sorted(e for e, _ in errs)[len(errs) // 2], cos=cos)
        print(f'[bf16 VQ-AE step vs oracle autograd, {label}] recon rel {results[label]["recon"]:.3e}, loss rel '
              f'{results[label]["loss"]:.3e}, dx rel {results[label]["dx"]:.3e}, gradients: median {results[label]["median"]:.3e}, '
              f'worst {worst[0]:.3e} ({worst[1]}), min cosine {cos:.4f}')
    # What bounds a bf16 gradient here: an element whose pre-activation lies within the bf16 error of zero changes its LeakyReLU
    # mask, i.e. its whole contribution (~0.4 % of the elements per layer => ~6 % per normalised layer against a non-smooth
    # upstream gradient, compounding over the eight BatchNorm + LeakyReLU stages in front of the first encoder block:
    # tools/diag_resblock_grad.py, test_zero_inserted_plane_of_a_strided_data_gradient pins it to the mask on one layer; the fp32
    # route's 5e-4 in test_autoencoder_gpu.py is the exact check of the same kernels' arithmetic).  Forward quantities are tight.
    r = results['same assignment']
    assert r['recon'] < 3e-2 and r['loss'] < 5e-3, r
    assert r['dx'] < 0.4 and r['worst'][0] < 0.4 and r['median'] < 0.15 and r['cos'] > 0.9, r
    # (against the oracle's OWN assignment the 3 % of flipped near-tie codes each move a 4 x 4 patch of the reconstruction: reported,
    #  only the loss is bounded)
    assert results['oracle assignment']['loss'] < 3e-2, results['oracle assignment']


@pytest.mark.parametrize('geom', [(4, 16, 16, 128, 64, 1, 1, 0, True), (2, 32, 32, 64, 64, 2, 2, 0, False), (4, 16, 16, 8, 64, 3, 1, 1, False),
                                  (1, 8, 8, 64, 128, 1, 1, 0, False), (2, 16, 16, 128, 128, 1, 1, 0, True), (4, 16, 16, 8, 128, 3, 1, 1, False),
                                  (2, 8, 16, 64, 24, 1, 1, 0, True), (3, 16, 16, 64, 64, 2, 2, 0, False), (4, 24, 24, 16, 40, 3, 2, 1, False),
                                  (16, 64, 64, 128, 64, 1, 1, 0, True)])
def test_small_k_conv_vs_torch(geom):
    """csrc/conv_point.hip at its own shapes against torch F.conv2d on the same bf16-representable operands (fp32 arithmetic):
    an EXTERNAL reference (the bit-comparison with conv2d_kernel in test_autoencoder_gpu.py is HIP against HIP) -- prologue
    (BatchNorm apply + LeakyReLU on load), bias, folded affine, LeakyReLU, channel statistics."""
    from world_modelz_amd import ops
    F = torch.nn.functional
    B, H, W, Ci, Co, k, st, pad, pre = geom
    assert ops.L.lib().wmz_conv_point_supported(B, H, W, Ci, Co, k, k, st, pad)
    torch.manual_seed(6)
    x = torch.randn(B, H, W, Ci, device='cuda').bfloat16()
    w = (torch.randn(Co, k * k * Ci, device='cuda') * (k * k * Ci) ** -0.5).bfloat16()
    bias = torch.randn(Co, device='cuda')
    sc, sh = torch.rand(Co, device='cuda') + 0.5, torch.randn(Co, device='cuda')
    prol = (torch.rand(Ci, device='cuda') + 0.5, torch.randn(Ci, device='cuda') * 0.3, 0.01) if pre else None
    xin = x.float()
    if pre:
        xin = F.leaky_relu(xin * prol[0] + prol[1], 0.01).bfloat16().float()          # (the kernel rounds the prologue's result to the MFMA operand)
    w4 = w.float().view(Co, k, k, Ci).permute(0, 3, 1, 2)
    base = F.conv2d(xin.permute(0, 3, 1, 2), w4, None, stride=st, padding=pad).permute(0, 2, 3, 1)
    for kw, ref in ((dict(bias=bias, leaky=True, stats=True), F.leaky_relu(base + bias, 0.01)), (dict(stats=True), base),
                    (dict(scale=sc, shift=sh, leaky=True), F.leaky_relu(base * sc + sh, 0.01)),
                    (dict(bias=bias, scale=sc, shift=sh, stats=True), (base + bias) * sc + sh)):
        with recorded_calls() as seen:
            out = ops.conv2d_nhwc(x, w, k, k, st, pad, pre=prol, **kw)
        assert seen[-1] == 'wmz_conv_point_fwd_bn', seen
        y = out[0] if kw.get('stats') else out
        assert y.shape == ref.shape
        err = float((y.float() - ref).norm() / ref.norm())
        assert err < 3e-3, (geom, list(kw), err)                                      # (one bf16 rounding of the stored result)
        assert float((y.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
        if kw.get('stats'):
            M = y.numel() // Co
            assert torch.allclose(out[1].sum(0) / M, ref.mean((0, 1, 2)), rtol=5e-3, atol=2e-3)
            assert torch.allclose(out[2].sum(0) / M, (ref ** 2).mean((0, 1, 2)), rtol=5e-3, atol=2e-3)


def test_wide_hidden_planes_keep_the_implicit_gemm_prologue():
    """a Residual block with hidden_planes in 136..256 (the reference's argparse accepts --hidden_planes 256,
    train_vqae.py:206) reaches the 1x1 conv with a BatchNorm prologue over MORE than 128 channels: the streaming kernel's prologue is
    not built for that, so the layer stays on the implicit-GEMM kernel (it used to raise).  Latents against the oracle."""
    from world_modelz_amd import config
    from world_modelz_amd.train_vqae import VqAutoEncoder
    torch.manual_seed(5)
    m = VqAutoEncoder(embedding_dim=64, num_embeddings=128, downscale_steps=1, hidden_planes=256).cuda()
    m.train()
    sd = _cpu_state(m)
    frames = torch.rand(8, 3, 32, 32)
    lat_ref = oae.encoder_forward({k: v.clone() for k, v in sd.items()}, frames, training=True).permute(0, 2, 3, 1)
    with config.compute_dtype(torch.bfloat16), torch.no_grad(), recorded_calls() as seen:
        lat = m._latents(frames.cuda())
    assert 'wmz_conv2d_nhwc_fwd_pre' in seen
    assert rel(lat, lat_ref) < 2e-2, rel(lat, lat_ref)


def test_channel_widths_off_the_granule_raise():
    from world_modelz_amd._lib import WmzError
    from world_modelz_amd.train_vqae import VqAutoEncoder
    with pytest.raises(WmzError):
        VqAutoEncoder(embedding_dim=20, num_embeddings=32, downscale_steps=1, hidden_planes=24)
    with pytest.raises(WmzError):
        VqAutoEncoder(embedding_dim=16, num_embeddings=32, downscale_steps=1, hidden_planes=30)
