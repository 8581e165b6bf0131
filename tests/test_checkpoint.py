"""Reference checkpoint format: the two `.pth` fixtures under tests/golden/ were written by
tests/golden/make_golden.py with the reference's own argparse parsers, model classes, ModelEmaV2 and torch.save dicts
(main.py:297-309, train_vqae.py:168-179)."""
import argparse
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, near_tie_mismatches

DEN = os.path.join(GOLDEN, 'ckpt_denoiser_tiny.pth')
VQAE = os.path.join(GOLDEN, 'ckpt_vqae_tiny.pth')


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_checkpoint_files_have_the_reference_layout():
    """CPU: keys, the pickled Namespace and the state_dict schema -- no compute."""
    from world_modelz_amd import checkpoint
    d = checkpoint.read(DEN)
    assert set(d) == {'step', 'lr', 'model_state_dict', 'ema_model_state_dict', 'optimizer_state_dict', 'opt'}
    assert isinstance(d['opt'], argparse.Namespace) and d['opt'].extents == '1,1,1' and d['opt'].n_past == 2
    assert d['opt'].decoder_model == 'ckpt_vqae_tiny.pth'
    a = checkpoint.read(VQAE)
    assert set(a) == {'step', 'lr', 'model_state_dict', 'optimizer_state_dict', 'loss', 'opt'}
    assert a['opt'].num_embeddings == 16 and a['opt'].embedding_dim == 8
    assert 'vq.embedding' in a['model_state_dict'] and 'vq.activation_count' not in a['model_state_dict']
    assert set(d['model_state_dict']) == set(d['ema_model_state_dict'])


@pytest.mark.gpu
@pytest.mark.parametrize('use_ema', [False, True])
def test_load_reference_checkpoint_reproduces_the_reference(use_ema):
    """main.py:365-410 through load_reference_checkpoint: the AE is found through the path recorded in the pickled opt,
    both models rebuild from the Namespaces alone, frames -> tokens -> logits match what the reference computed."""
    from world_modelz_amd import checkpoint, config
    g = load_golden('ckpt_tiny_expect')
    model, ae, opt, ae_opt = checkpoint.load_reference_checkpoint(DEN, use_ema=use_ema)
    assert ae.training and ae_opt.num_embeddings == ae.vq.num_embeddings == 16
    assert tuple(int(s) for s in g['data_shape']) == (model.transformer.pos_emb_s.num_embeddings,
                                                      model.transformer.pos_emb_h.num_embeddings,
                                                      model.transformer.pos_emb_w.num_embeddings)
    with config.compute_dtype(torch.float32), torch.no_grad():
        tokens = ae.encode(g['frames'].cuda()).view(2, 3, 4, 4)
        recon = ae.decode(g['tokens'].cuda().view(-1, 4, 4))
        logits = model(g['tokens'].cuda())
    assert rel(recon, g['recon']) < 1e-5
    assert rel(logits, g['logits_ema' if use_ema else 'logits']) < 1e-5
    assert not use_ema or rel(logits, g['logits']) > 1e-4            # the EMA weights really are different weights
    if not torch.equal(tokens.cpu(), g['tokens']):                   # only a genuine near-tie may differ
        from oracle import autoencoder as oae
        sd = {k: v.clone() for k, v in checkpoint.read(VQAE)['model_state_dict'].items()}
        lat = oae.encoder_forward(sd, g['frames'], training=True).permute(0, 2, 3, 1)
        near_tie_mismatches(tokens, g['tokens'], lat, sd['vq.embedding'][0])


@pytest.mark.gpu
def test_saved_checkpoints_round_trip(tmp_path):
    """save_*_checkpoint writes the reference's dicts: reading them back through the reference-format loader gives
    bit-identical weights and the same Namespace."""
    from world_modelz_amd import checkpoint
    model, ae, opt, ae_opt = checkpoint.load_reference_checkpoint(DEN)
    pa, pd = str(tmp_path / 'ae.pth'), str(tmp_path / 'den.pth')
    checkpoint.save_vqae_checkpoint(pa, step=7, lr=[1e-4], model=ae, opt=ae_opt)
    opt.decoder_model = pa
    checkpoint.save_denoiser_checkpoint(pd, step=7, lr=1e-4, model=model, opt=opt, ema_model=model)
    m2, a2, o2, _ = checkpoint.load_reference_checkpoint(pd, use_ema=True)
    assert vars(o2) == vars(opt)
    for (k, v), (k2, v2) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    for (k, v), (k2, v2) in zip(ae.state_dict().items(), a2.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
