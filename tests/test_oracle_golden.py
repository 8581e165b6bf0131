"""CPU: the oracle (oracle/) against every golden vector captured from the reference
(tests/golden/make_golden.py).  This is what pins the oracle; the -m gpu tests then compare the
HIP path with the oracle and with the same vectors."""
import math

import pytest
import torch

from conftest import load_golden, sub
from oracle import attention as oat
from oracle import autoencoder as oae
from oracle import denoiser as oden
from oracle import train_step as ots
from oracle import vq as ovq


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('tag', list('abcdef'))
def test_attention_core(tag):
    g = load_golden(f'attn_core_{tag}')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    out, logits = oat.local_attention(g['k'], g['v'], g['q'], ext, heads, return_logits=True)
    # masked slots carry the literal -1e9 (quirk Q2)
    assert torch.equal(logits == oat.MASK_VALUE, g['logits'] == -1e9)
    live = g['logits'] != -1e9
    assert torch.allclose(logits[live], g['logits'][live], rtol=1e-5, atol=2e-6)
    assert rel(out, g['out']) < 2e-6


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_attention_module_and_grads(tag):
    g = load_golden(f'attn_module_{tag}')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    sd = {k: v.clone().requires_grad_(True) for k, v in sub(g, 'sd/').items()}
    x = g['x'].clone().requires_grad_(True)
    q = g['q'].clone().requires_grad_(True)
    out = oat.attention_module(sd, '', x, q, ext, heads)
    assert rel(out, g['out']) < 2e-6
    out.square().sum().backward()
    assert rel(x.grad, g['dx']) < 1e-5
    assert rel(q.grad, g['dq']) < 1e-5
    for k, v in sub(g, 'grad/').items():
        assert rel(sd[k].grad, v) < 1e-5, k


def test_transformer_tiny():
    g = load_golden('transformer_tiny')
    sd = sub(g, 'sd/')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    x = oden.transformer_forward(sd, g['z'], ext, heads)
    assert rel(x, g['x_final']) < 2e-6
    assert rel(oden.denoiser_forward(sd, g['z'], ext, heads), g['logits']) < 2e-6
    # shorter clip than the position table (reference test(), local_3d_attention.py:168-171)
    assert rel(oden.denoiser_forward(sd, g['z_short'], ext, heads), g['logits_short']) < 2e-6
    # batch chunking is exact (clips are independent)
    assert torch.equal(oden.denoiser_forward(sd, g['z'], ext, heads, batch_chunk=1),
                       oden.denoiser_forward(sd, g['z'], ext, heads))


def test_transformer_identity_out():
    g = load_golden('transformer_identity_out')
    sd = sub(g, 'sd/')
    assert 'transformer.layers.0.0.fn.to_out.0.weight' not in sd      # quirk Q6
    ext = tuple(int(e) for e in g['extents'])
    assert rel(oden.denoiser_forward(sd, g['z'], ext, int(g['heads'])), g['logits']) < 2e-6


@pytest.mark.parametrize('C', [512, 1024, 8192])
def test_vq_encode(C):
    g = load_golden(f'vq_encode_{C}')
    idx = ovq.encode(g['x'], g['embedding'])
    assert idx.dtype == torch.int64 and torch.equal(idx, g['idx'])           # bit-identical
    assert int(idx[5, 0]) == 7 and int(idx[6, 0]) == 7                       # tie -> lowest index
    d = ovq.distances(g['x'], g['embedding'])[:, 0]
    assert torch.equal(d[:8], g['dist_rows'])
    assert torch.equal(d[:8] / 64, g['dist_rows_normalized'])
    assert torch.equal(d.min(dim=-1).values, g['dist_min'])
    assert torch.equal(ovq.decode(idx, g['embedding']), g['decoded'])
    # the explicit 8-lane x 4-accumulator order is what ATen evaluates on x86 (bitwise)
    n = 64
    assert torch.equal(ovq.distances_avx_order(g['x'][:n], g['embedding'])[:, 0], d[:n])


def test_vq_encode_odd():
    g = load_golden('vq_encode_odd')
    assert torch.equal(ovq.encode(g['x'], g['embedding']), g['idx'])
    assert torch.equal(ovq.distances(g['x'], g['embedding'])[:, 0], g['dist'])
    assert torch.equal(ovq.distances_avx_order(g['x'], g['embedding'])[:, 0], g['dist'])


def test_vq_forward_sequence():
    g = load_golden('vq_forward_train')
    st = {'embedding': g['embedding0'].clone(), 'cluster_size': g['cluster_size0'].clone(),
          'activation_count': torch.zeros(1, 32), 'accumulated_error': torch.zeros(1, 32)}
    for tag, training in [('t0', True), ('t1', True), ('t2', True), ('e', False)]:
        x = g[f'{tag}/x'].clone().requires_grad_(training)
        qz, enc, loss, ppl = ovq.forward(x, st, training)
        assert torch.equal(enc.argmax(-1), g[f'{tag}/encodings_argmax'])
        assert torch.allclose(qz, g[f'{tag}/quantized'], rtol=0, atol=1e-6)
        assert torch.allclose(loss, g[f'{tag}/loss'], rtol=1e-6)
        assert torch.allclose(ppl, g[f'{tag}/perplexity'], rtol=1e-6)
        for b in st:
            assert torch.allclose(st[b], g[f'{tag}/{b}'], rtol=1e-6, atol=1e-7), (tag, b)
    assert ovq.reuse_inactive(st) == int(g['reused'])
    for b in st:
        assert torch.allclose(st[b], g[f'reuse/{b}'], rtol=1e-6, atol=1e-7)
    ovq.reset_stats(st)
    for b in st:
        assert torch.allclose(st[b], g[f'reset/{b}'], rtol=1e-6, atol=1e-7)


def test_autoencoder_roundtrip():
    g = load_golden('ae_roundtrip')
    p = oae.with_vq_stats({k: v.clone() for k, v in sub(g, 'sd0/').items()})
    x = g['x']
    h = oae.encoder_forward(p, x, training=False)
    assert rel(h, g['eval/enc_out']) < 2e-6
    idx = oae.vqae_encode(p, x, training=False)
    assert torch.equal(idx, g['eval/idx'])
    assert rel(oae.vqae_decode(p, idx, training=False), g['eval/decoded']) < 2e-6
    rec, ll, ppl = oae.vqae_forward(p, x, training=False)
    assert rel(rec, g['eval/recon']) < 2e-6
    assert torch.allclose(ll, g['eval/latent_loss'], rtol=1e-5)
    assert torch.allclose(ppl, g['eval/perplexity'], rtol=1e-5)
    # quirk Q3: BatchNorm in training mode under no_grad changes the indices and the running stats
    idx_t = oae.vqae_encode(p, x, training=True)
    assert torch.equal(idx_t, g['train/idx'])
    assert not torch.equal(idx_t, idx)
    for k, v in sub(g, 'sd1/').items():
        if v.dtype.is_floating_point:
            assert torch.allclose(p[k], v, rtol=1e-5, atol=1e-6), k
        else:
            assert torch.equal(p[k], v), k
    rec, ll, ppl = oae.vqae_forward(p, x, training=True)
    assert rel(rec, g['train/recon']) < 5e-6
    assert torch.allclose(ll, g['train/latent_loss'], rtol=1e-5)
    for k, v in sub(g, 'sd2/').items():
        if v.dtype.is_floating_point:
            assert torch.allclose(p[k], v, rtol=1e-5, atol=1e-6), k


def test_autoencoder_autograd_vs_reference_capture():
    """The oracle's VqAutoEncoder.forward under torch autograd (straight-through estimator and commitment loss of vq.py:67-70,
    EMA update outside the graph) reproduces the reference's captured gradients: the GPU tests compare the bf16 conv route's
    gradients with THIS autograd, so it is pinned here first."""
    g = load_golden('ae_roundtrip')
    sd1 = sub(g, 'sd1/')
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k and not k.startswith('vq.')
                  else v.clone()) for k, v in sd1.items()}
    x = g['x'].clone().requires_grad_(True)
    rec, ll, _ = oae.vqae_forward(oae.with_vq_stats(leaves), x, training=True)
    loss = torch.nn.functional.smooth_l1_loss(rec, g['x']) + 0.25 * ll
    loss.backward()
    assert torch.allclose(loss.detach(), g['train/loss'], rtol=1e-5)
    assert rel(x.grad, g['train/dx']) < 2e-5
    names = [k[len('train/grad/'):] for k in g if k.startswith('train/grad/')]
    assert len(names) > 40
    floor = 1e-3 * max(float(g['train/grad/' + n].norm()) for n in names)      # (biases in front of a training-mode BatchNorm: true gradient 0)
    for n in names:
        ref = g['train/grad/' + n]
        assert leaves[n].grad is not None, n
        assert float((leaves[n].grad - ref).norm()) / max(float(ref.norm()), floor) < 5e-5, n
    # the same forward with the assignment handed in (the test knob the GPU gradient tests use) is the same computation
    leaves2 = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k and not k.startswith('vq.')
                   else v.clone()) for k, v in sd1.items()}
    rec2, ll2, _ = oae.vqae_forward(oae.with_vq_stats(leaves2), g['x'], training=True, assign=g['train/idx'].reshape(-1, 1))
    assert torch.equal(rec2, rec) and torch.equal(ll2, ll)


def test_training_step():
    g = load_golden('step_tiny')
    sd0 = sub(g, 'sd0/')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    C = g['logits'].shape[-1]
    zc, target = ots.corrupt_last_frame(g['batch_z'], g['r'], g['mask_uniform'], g['draw'], C)
    assert torch.equal(zc, g['corrupted']) and torch.equal(target, g['target'])
    assert torch.allclose(ots.corruption_probs(g['batch_z'][:, -1], g['r'], C), g['d_probs'], rtol=1e-6)
    y, per_sample, loss, grads = ots.step_grads(sd0, zc, target, ext, heads)
    assert rel(y, g['logits']) < 2e-6
    assert torch.allclose(per_sample, g['per_sample_loss'], rtol=1e-5)
    assert torch.allclose(loss, g['loss'], rtol=1e-5)
    for k, v in sub(g, 'grad/').items():
        assert rel(grads[k], v) < 2e-5, k
    assert math.isclose(ots.grad_norm(grads), float(g['grad_norm']), rel_tol=1e-5)
    # one AdamW step from sd0
    p = {k: v.clone() for k, v in sd0.items()}
    ots.adamw_step(p, grads, {}, lr=float(g['adamw_lr']))
    for k, v in sub(g, 'sd1/').items():
        assert torch.allclose(p[k], v, rtol=1e-5, atol=1e-7), k
    # warm-up + cosine learning-rate trajectory
    traj = g['lr_trajectory'].tolist()
    mine = [ots.lr_at(s, float(g['base_lr']), int(g['warmup']), int(g['max_steps'])) for s in range(1, len(traj) + 1)]
    assert all(math.isclose(a, b, rel_tol=1e-6, abs_tol=1e-12) for a, b in zip(mine, traj)), (mine, traj)
    assert math.isclose(ots.lr_at(4, float(g['base_lr']), 5, 1000), float(g['lr_used']), rel_tol=1e-6)


def test_resample_matches_lerp_distribution():
    """The closed form used by the fused corruption kernel has the reference's categorical law."""
    C, r = 16, torch.tensor([0.7])
    z = torch.tensor([[3]])
    d = ots.corruption_probs(z, r, C)[0, 0]
    a = 0.1 * 0.7
    closed = torch.full((C,), a / C)
    closed[3] += 1 - a
    assert torch.allclose(d, closed, atol=1e-7)
    torch.manual_seed(0)
    n = 200000
    draws = ots.resample_tokens(z.expand(1, n).reshape(1, -1), r, torch.rand(1, n), torch.rand(1, n), C)
    freq = torch.bincount(draws.view(-1), minlength=C).float() / n
    assert torch.allclose(freq, closed, atol=4e-3)


def test_loss_aware_sampler():
    g = load_golden('step_tiny')
    s = ots.LossAwareSampler(buckets=10, uniform_p=0.01, alpha=0.9, warmup=2)
    s.update(g['sampler/ts'], g['sampler/losses'])
    assert torch.equal(s.counts, g['sampler/counts'])
    assert torch.allclose(s.w, g['sampler/weights_raw'], rtol=1e-6)
    assert s.warmed_up() == bool(g['sampler/warmed_up'])
    assert torch.allclose(s.weights(), g['sampler/weights'], rtol=1e-6)


def test_sparse_tiny():
    g = load_golden('sparse_tiny')
    sd = sub(g, 'sd/')
    shape = tuple(int(e) for e in g['shape'])
    out = oden.sparse_denoiser_forward(sd, g['x'], g['indices'], shape, int(g['heads']))
    assert rel(out, g['logits']) < 2e-6


@pytest.mark.parametrize('E', [8, 16, 17, 20, 24, 32, 40, 56, 64, 72, 100, 128, 200, 256])
def test_vq_distance_summation_order(E):
    """ATen's reduction order for the reference expression (vq.py:30), restated explicitly -- this is the
    order the HIP argmin kernel implements, so it must be bitwise the torch result on this host."""
    torch.manual_seed(E)
    x, emb = torch.randn(37, E), torch.randn(1, 29, E)
    assert torch.equal(ovq.distances_avx_order(x, emb), ovq.distances(x, emb))


@pytest.mark.parametrize('name', ['sampler_tiny', 'sampler_tiny_topk'])
def test_sampler_loop_vs_reference_capture(name):
    """oracle.sampler.evaluate_tokens == the reference's evaluate_model (main.py:50-117) run with the same injected
    uniform fields: every last frame fed to the model and every generated frame, token for token."""
    from oracle import sampler as osamp
    g = load_golden(name)
    sd = sub(g, 'model/')
    ext, heads, C = tuple(int(e) for e in g['extents']), int(g['heads']), int(g['num_embeddings'])
    frames, z_final, fed = osamp.evaluate_tokens(lambda z: oden.denoiser_forward(sd, z, ext, heads), g['z0'], C,
                                                 g['tokens'].shape[0], g['u_multi'], g['u_mask'], sample_topk=int(g['topk']))
    assert torch.equal(torch.stack(fed), g['fed'])
    assert torch.equal(torch.stack(frames), g['tokens'])
    assert float(g['margin']) > 2e-5                       # every categorical draw sits clear of a CDF step
    # the frames shift by one per generated frame (:115): [c0, c1, *] -> [c1, f1, f1] -> [f1, f2, f2]
    assert torch.equal(z_final[:, 0], frames[0]) and torch.equal(z_final[:, 1], frames[1])
    assert torch.equal(z_final[:, 2], frames[1])
