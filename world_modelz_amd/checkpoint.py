"""Reference checkpoint format (SURVEY 5 "Checkpoint / resume", 8f N4): load the `.pth` files vq-video-diffusion writes
into the MI355X modules, and write files the reference scripts can load back.

On disk (torch.save of a dict):
  denoiser  main.py:302-309        {'step', 'lr', 'model_state_dict', 'ema_model_state_dict', 'optimizer_state_dict', 'opt'}
  VQ-AE     train_vqae.py:172-179  {'step', 'lr', 'model_state_dict', 'optimizer_state_dict', 'loss', 'opt'}
`opt` is the pickled argparse.Namespace of the writing script; the reference rebuilds its models from it:
the frozen VQ-AE from ITS checkpoint's opt.embedding_dim / num_embeddings / downscale_steps / hidden_planes with
in_channels = 1 (main.py:376-379), the denoiser from the denoiser checkpoint's opt.dim / extents / depth / mlp_dim /
dim_head / heads / dropout and the AE's codebook size (main.py:390-402).  Resume is weights-only in the reference
(main.py:409-410): optimizer / scheduler state is carried in the file but never restored, and so it is here.
"""
import argparse
import os

import torch

from .main import VqVideoDiffusionModel
from .train_vqae import VqAutoEncoder


def read(path, map_location='cpu'):
    """torch.load of a reference checkpoint.  The only non-tensor global in these files is argparse.Namespace, so the
    restricted (weights_only) unpickler is used with that one class allow-listed."""
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(path, map_location=map_location, weights_only=True)


def build_vqae(data, in_channels=1):
    """VqAutoEncoder rebuilt from a VQ-AE checkpoint dict exactly as main.py:376-379 does (in_channels is not stored in
    the file: main.py hard-codes 1, minecraft/main2.py 3)."""
    o = data['opt']
    model = VqAutoEncoder(o.embedding_dim, o.num_embeddings, o.downscale_steps, hidden_planes=o.hidden_planes,
                          in_channels=in_channels)
    model.load_state_dict(data['model_state_dict'], strict=True)
    return model


def build_denoiser(data, num_embeddings, use_ema=False):
    """VqVideoDiffusionModel rebuilt from a denoiser checkpoint dict (main.py:390-410).  The reference takes data_shape
    from an encoded sample (`z.shape`, :388-394); the position-embedding tables in the state_dict record that shape."""
    o = data['opt']
    sd = data['ema_model_state_dict'] if use_ema else data['model_state_dict']
    if sd is None:
        raise KeyError('the checkpoint holds no EMA weights (it was written with --ema_decay 0)')
    data_shape = tuple(sd[f'transformer.pos_emb_{a}.weight'].shape[0] for a in 'shw')
    extents = [int(e) for e in str(o.extents).split(',')]
    assert len(extents) == 3
    model = VqVideoDiffusionModel(data_shape=data_shape, dim=o.dim, num_classes=num_embeddings, extents=extents,
                                  depth=o.depth, mlp_dim=o.mlp_dim, dim_head=o.dim_head, heads=o.heads, dropout=o.dropout)
    model.load_state_dict(sd, strict=True)
    return model


def load_reference_checkpoint(checkpoint, decoder_model=None, device='cuda', use_ema=False, in_channels=1):
    """main.py:365-410 in one call: returns (denoiser, frozen VQ-AE, opt, AE opt), both models on `device`.

    checkpoint: a denoiser `.pth`; decoder_model: the VQ-AE `.pth` (default: the path recorded in the denoiser
    checkpoint's opt.decoder_model, looked up as given and then next to the denoiser file).  Like the reference, the AE
    is left in train mode (quirk Q3: its BatchNorm layers keep using batch statistics)."""
    data = read(checkpoint)
    opt = data['opt']
    if decoder_model is None:
        decoder_model = opt.decoder_model
        if not os.path.exists(decoder_model):
            decoder_model = os.path.join(os.path.dirname(os.path.abspath(checkpoint)), os.path.basename(decoder_model))
    ae_data = read(decoder_model)
    ae = build_vqae(ae_data, in_channels=in_channels)
    model = build_denoiser(data, ae_data['opt'].num_embeddings, use_ema=use_ema)
    return model.to(device), ae.to(device), opt, ae_data['opt']


def save_denoiser_checkpoint(path, *, step, lr, model, opt, ema_model=None, optimizer_state=None):
    """Write what main.py:302-309 writes (the reference's `--checkpoint` loader reads it back)."""
    torch.save({'step': int(step), 'lr': list(lr) if isinstance(lr, (list, tuple)) else [float(lr)],
                'model_state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                'ema_model_state_dict': (None if ema_model is None else
                                         {k: v.detach().cpu() for k, v in ema_model.state_dict().items()}),
                'optimizer_state_dict': optimizer_state, 'opt': opt}, path)


def save_vqae_checkpoint(path, *, step, lr, model, opt, optimizer_state=None, train_recon_error=()):
    """Write what train_vqae.py:172-179 writes (main.py's `--decoder_model` loader reads it back)."""
    torch.save({'step': int(step), 'lr': list(lr) if isinstance(lr, (list, tuple)) else [float(lr)],
                'model_state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                'optimizer_state_dict': optimizer_state, 'loss': {'train_recon_error': list(train_recon_error)},
                'opt': opt}, path)
