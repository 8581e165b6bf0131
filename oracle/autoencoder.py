"""Oracle (test infrastructure): conv encoder / decoder and VqAutoEncoder, fp32 CPU.

Restates vq-video-diffusion/autoencoder.py:8-152 and train_vqae.py:22-55 as functions over a
state_dict (schema: SURVEY.md appendix A).  BatchNorm follows nn.BatchNorm2d defaults
(eps 1e-5, momentum 0.1); with training=True it normalises with batch statistics and updates
running_mean / running_var (unbiased) / num_batches_tracked in `params` in place -- this is
what the frozen AE does inside main.py because it is never .eval()-ed (quirk Q3,
main.py:234-237, :378-379).
"""
import torch
import torch.nn.functional as F

from . import vq as ovq

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY = 0.01  # nn.LeakyReLU / F.leaky_relu default negative slope


def batch_norm(params, prefix, x, training):
    rm, rv = params[prefix + 'running_mean'], params[prefix + 'running_var']
    y = F.batch_norm(x, rm, rv, params[prefix + 'weight'], params[prefix + 'bias'],
                     training, BN_MOMENTUM, BN_EPS)
    if training and prefix + 'num_batches_tracked' in params:
        params[prefix + 'num_batches_tracked'] += 1
    return y


def residual_block(params, prefix, x, stride, training):
    """Residual (autoencoder.py:18-42)."""
    h = F.conv2d(x, params[prefix + '_block.0.weight'], None, stride=stride, padding=1)
    h = F.leaky_relu(batch_norm(params, prefix + '_block.1.', h, training), LEAKY)
    h = F.conv2d(h, params[prefix + '_block.3.weight'])
    h = batch_norm(params, prefix + '_block.4.', h, training)
    if stride != 1:
        r = F.conv2d(x, params[prefix + 'downsample.0.weight'], None, stride=stride)
        r = batch_norm(params, prefix + 'downsample.1.', r, training)
    else:
        r = x
    return F.leaky_relu(h + r, LEAKY)


def num_downscale_steps(params, prefix='encoder.'):
    n = 0
    while f'{prefix}_residual_stack._stack.{2 * n}._block.0.weight' in params:
        n += 1
    return n


def encoder_forward(params, x, training, prefix='encoder.'):
    """SimpleResidualEncoder.forward (autoencoder.py:83-86): NCHW in, NCHW out."""
    h = F.leaky_relu(F.conv2d(x, params[prefix + '_conv_1.weight'], None, padding=1), LEAKY)
    for n in range(num_downscale_steps(params, prefix)):
        h = residual_block(params, f'{prefix}_residual_stack._stack.{2 * n}.', h, 1, training)
        h = residual_block(params, f'{prefix}_residual_stack._stack.{2 * n + 1}.', h, 2, training)
    return h


def upscale_residual(params, prefix, x, training):
    """UpscaleResidual.forward with bilinear x2, align_corners=False (autoencoder.py:120-133, :138)."""
    def up(t):
        return F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False)
    h = F.leaky_relu(batch_norm(params, prefix + 'bn1.', x, training), LEAKY)
    h = up(h)
    x = up(x)
    h = F.conv2d(h, params[prefix + 'conv1.weight'], params[prefix + 'conv1.bias'], padding=1)
    h = F.leaky_relu(batch_norm(params, prefix + 'bn2.', h, training), LEAKY)
    h = F.conv2d(h, params[prefix + 'conv2.weight'], params[prefix + 'conv2.bias'], padding=1)
    x = F.conv2d(x, params[prefix + 'conv_residual.weight'], params[prefix + 'conv_residual.bias'])
    return h + x


def decoder_forward(params, x, training, prefix='decoder.'):
    """SimpleResidualDecoder.forward (autoencoder.py:134-152)."""
    h = F.conv2d(x, params[prefix + 'decoder_stack.0.weight'], None, padding=1)
    u = 1
    while f'{prefix}decoder_stack.{u}.conv1.weight' in params:
        h = upscale_residual(params, f'{prefix}decoder_stack.{u}.', h, training)
        u += 1
    return F.conv2d(h, params[f'{prefix}decoder_stack.{u}.weight'], None, padding=1)


def _vq_state(params):
    return {'embedding': params['vq.embedding'], 'cluster_size': params['vq.cluster_size'],
            'activation_count': params['vq.activation_count'],
            'accumulated_error': params['vq.accumulated_error']}


def with_vq_stats(params):
    """state_dict lacks the non-persistent buffers (vq.py:18-20); add zeroed ones."""
    p = dict(params)
    L, C, _ = p['vq.embedding'].shape
    p.setdefault('vq.activation_count', torch.zeros(L, C))
    p.setdefault('vq.accumulated_error', torch.zeros(L, C))
    return p


def vqae_encode(params, x, training):
    """VqAutoEncoder.encode (train_vqae.py:45-49): frames NCHW -> int64 [B,h,w]."""
    h = encoder_forward(params, x, training).permute(0, 2, 3, 1)
    return ovq.encode(h, params['vq.embedding']).view(h.shape[:-1])


def vqae_decode(params, z, training):
    """VqAutoEncoder.decode (train_vqae.py:51-55)."""
    h = ovq.decode(z, params['vq.embedding']).permute(0, 3, 1, 2)
    return decoder_forward(params, h, training)


def vqae_forward(params, x, training, assign=None):
    """VqAutoEncoder.forward (train_vqae.py:33-43): (recon, latent_loss, perplexity).  assign: oracle.vq.forward's test knob."""
    h = encoder_forward(params, x, training).permute(0, 2, 3, 1)
    q, _, loss, ppl = ovq.forward(h, _vq_state(params), training, assign=assign)
    q = q.permute(0, 3, 1, 2).contiguous()
    return decoder_forward(params, q, training), loss, ppl
