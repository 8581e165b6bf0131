"""The zero-edit drop-in: the reference scripts import their model code by BARE module name from sibling
files (`from local_3d_attention import Local3dAttentionTransformer`, main.py:19; `from autoencoder import ...`,
`from vq import VectorQuantizerEMA`, train_vqae.py:18-19).  With world_modelz_amd/dropin/ in front of sys.path those
imports resolve to the MI355X classes, and the reference's own model classes -- restated here exactly as they compose
the modules -- must work unchanged, in the library-default compute dtype (bf16) AND in fp32:

  * VqVideoDiffusionModel (main.py:25-36): transformer -> x[:, -1] -> a plain torch fp32 nn.Linear;
  * VqAutoEncoder (train_vqae.py:22-55): encoder -> permute(0,2,3,1) -> vq(...) -> permute(0,3,1,2).contiguous()
    -> decoder, and encode / decode.
"""
import importlib
import os
import sys

import pytest
import torch
from torch import nn

from conftest import ROOT, load_golden, near_tie_mismatches, sub

pytestmark = pytest.mark.gpu

DROPIN = os.path.join(ROOT, 'world_modelz_amd', 'dropin')
BARE = ('local_3d_attention', 'vq', 'autoencoder')


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def bare():
    """Import the three modules by bare name through the shim directory, like the reference scripts do."""
    assert torch.cuda.is_available()
    saved = {k: sys.modules.pop(k, None) for k in BARE}
    sys.path.insert(0, DROPIN)
    try:
        mods = {k: importlib.import_module(k) for k in BARE}
        for k, m in mods.items():
            assert os.path.dirname(os.path.abspath(m.__file__)) == DROPIN, (k, m.__file__)
        yield mods
    finally:
        sys.path.remove(DROPIN)
        for k in BARE:
            sys.modules.pop(k, None)
            if saved[k] is not None:
                sys.modules[k] = saved[k]


def _ref_style_denoiser(bare, **kw):
    Local3dAttentionTransformer = bare['local_3d_attention'].Local3dAttentionTransformer

    class VqVideoDiffusionModel(nn.Module):           # composition of main.py:25-36 (torch nn.Linear head, x[:, -1])
        def __init__(self, *, data_shape, dim, num_classes, extents, depth, dim_head, mlp_dim, heads=1, dropout=.0):
            super().__init__()
            self.transformer = Local3dAttentionTransformer(data_shape=data_shape, dim=dim, num_classes=num_classes + 1,
                                                           extents=extents, depth=depth, heads=heads, dim_head=dim_head,
                                                           mlp_dim=mlp_dim, dropout=dropout)
            self.logit_proj = nn.Linear(dim, num_classes)

        def forward(self, x):
            x = self.transformer(x)
            return self.logit_proj(x[:, -1])
    return VqVideoDiffusionModel(**kw)


def _ref_style_autoencoder(bare, embedding_dim, num_embeddings, downscale_steps, hidden_planes, in_channels):
    A, V = bare['autoencoder'], bare['vq']

    class VqAutoEncoder(nn.Module):                    # composition of train_vqae.py:22-55 (permute glue in torch)
        def __init__(self):
            super().__init__()
            self.encoder = A.SimpleResidualEncoder(in_channels, embedding_dim, downscale_steps, hidden_planes)
            self.decoder = A.SimpleResidualDecoder([hidden_planes] * downscale_steps, in_channels=embedding_dim,
                                                   out_channels=in_channels)
            self.vq = V.VectorQuantizerEMA(embedding_dim, num_embeddings)

        def forward(self, x):
            h = self.encoder(x)
            h = h.permute(0, 2, 3, 1)
            h, _, latent_loss, perplexity = self.vq(h)
            h = h.permute(0, 3, 1, 2).contiguous()
            return self.decoder(h), latent_loss, perplexity

        def encode(self, x):
            h = self.encoder(x).permute(0, 2, 3, 1)
            return self.vq.encode(h).view(h.shape[:-1])

        def decode(self, z):
            return self.decoder(self.vq.decode(z).permute(0, 3, 1, 2))
    return VqAutoEncoder()


@pytest.mark.parametrize('dtype,tol', [(None, 1.5e-2), (torch.float32, 1e-5)])
def test_reference_style_denoiser_through_dropin(bare, dtype, tol):
    from world_modelz_amd import config
    g = load_golden('transformer_tiny')
    sd = sub(g, 'sd/')
    ext = tuple(int(e) for e in g['extents'])
    m = _ref_style_denoiser(bare, data_shape=(4, 5, 6), dim=32, num_classes=50, extents=ext, depth=2, dim_head=16,
                            mlp_dim=48, heads=int(g['heads']))
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    dt = dtype if dtype is not None else config.get_compute_dtype()      # None: the library default (bf16)
    with config.compute_dtype(dt), torch.no_grad():
        h = m.transformer(g['z'].cuda())
        logits = m(g['z'].cuda())
        short = m(g['z_short'].cuda())
    assert h.dtype == torch.float32 and logits.dtype == torch.float32       # the boundary dtype rule
    e = (rel(h, g['x_final']), rel(logits, g['logits']), rel(short, g['logits_short']))
    print(f'[dropin denoiser, {dt}] rel errors {e}')
    assert max(e) < tol


@pytest.mark.parametrize('dtype,tol', [(None, 2e-2), (torch.float32, 5e-5)])
def test_reference_style_denoiser_training_step_through_dropin(bare, dtype, tol):
    """loss.backward() through the torch head into the HIP backward; every gradient against the reference capture."""
    from world_modelz_amd import config
    g = load_golden('step_tiny')
    sd = sub(g, 'sd0/')
    ext = tuple(int(e) for e in g['extents'])
    C = sd['logit_proj.weight'].shape[0]
    m = _ref_style_denoiser(bare, data_shape=(3, 4, 4), dim=16, num_classes=C, extents=ext, depth=2, dim_head=8,
                            mlp_dim=24, heads=int(g['heads']))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    dt = dtype if dtype is not None else config.get_compute_dtype()
    with config.compute_dtype(dt):
        y = m(g['corrupted'].cuda())
        loss = nn.functional.cross_entropy(y.reshape(-1, C), g['target'].reshape(-1).cuda(), reduction='none')
        loss.mean().backward()
    assert rel(y, g['logits']) < tol
    worst = max(rel(p.grad, g['grad/' + n]) for n, p in m.named_parameters())
    print(f'[dropin training step, {dt}] worst gradient rel error {worst:.2e}')
    assert worst < (tol if dtype is not None else 6e-2)


@pytest.mark.parametrize('dtype', [None, torch.float32])
def test_reference_style_autoencoder_through_dropin(bare, dtype):
    from world_modelz_amd import config
    g = load_golden('ae_roundtrip')
    m = _ref_style_autoencoder(bare, 16, 32, 2, 24, 3)
    m.load_state_dict(sub(g, 'sd0/'), strict=True)
    m = m.cuda().eval()
    x = g['x'].cuda()
    dt = dtype if dtype is not None else config.get_compute_dtype()
    with config.compute_dtype(dt), torch.no_grad():
        h = m.encoder(x)
        idx = m.encode(x)
        rec = m.decode(g['eval/idx'].cuda())
        out, ll, ppl = m(x)
    assert h.dtype == torch.float32 and rec.dtype == torch.float32 and out.dtype == torch.float32
    assert h.shape == g['eval/enc_out'].shape and idx.shape == g['eval/idx'].shape and idx.dtype == torch.int64
    if dt == torch.float32:
        assert rel(h, g['eval/enc_out']) < 1e-5 and rel(rec, g['eval/decoded']) < 1e-5
        n_bad = near_tie_mismatches(idx, g['eval/idx'], g['eval/enc_out'].permute(0, 2, 3, 1), g['sd0/vq.embedding'][0])
        if n_bad == 0:
            assert rel(out, g['eval/recon']) < 1e-5
            assert torch.allclose(ll.cpu(), g['eval/latent_loss'], rtol=1e-4)
    else:
        agree = float((idx.cpu() == g['eval/idx']).float().mean())
        print(f'[dropin AE, bf16] enc rel {rel(h, g["eval/enc_out"]):.2e}, decode rel {rel(rec, g["eval/decoded"]):.2e}, '
              f'index agreement {agree:.3f}')
        assert rel(h, g['eval/enc_out']) < 2e-2 and rel(rec, g['eval/decoded']) < 2e-2 and agree >= 0.9


def test_reference_style_autoencoder_training_through_dropin(bare):
    """recon + latent loss backward (train_vqae.py:145-150) through the permute glue; gradients vs the reference."""
    from world_modelz_amd import config
    g = load_golden('ae_roundtrip')
    m = _ref_style_autoencoder(bare, 16, 32, 2, 24, 3)
    m.load_state_dict(sub(g, 'sd1/'), strict=True)
    m = m.cuda().train()
    x = g['x'].cuda()
    with config.compute_dtype(torch.float32):
        out, ll, _ = m(x)
        loss = nn.functional.smooth_l1_loss(out, x) + 0.25 * ll
        loss.backward()
    assert out.dtype == torch.float32
    if rel(out, g['train/recon']) < 1e-4:              # no index flipped at a near-tie
        assert abs(float(loss) - float(g['train/loss'])) < 1e-5
        # a bias in front of a training-mode BatchNorm has an exactly-zero true gradient (1e-9 noise in the reference):
        # errors are measured against the typical gradient magnitude
        floor = 1e-3 * max(float(g['train/grad/' + n].norm()) for n, _ in m.named_parameters())
        worst = max(float((p.grad.cpu() - g['train/grad/' + n]).norm() / max(float(g['train/grad/' + n].norm()), floor))
                    for n, p in m.named_parameters())
        print(f'[dropin AE training] worst gradient rel error {worst:.2e}')
        assert worst < 5e-4
