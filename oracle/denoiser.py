"""Oracle (test infrastructure): the denoiser stack, fp32 CPU.

Restates vq-video-diffusion/local_3d_attention.py:11-31 (PreNorm, FeedForward),
:121-163 (Local3dAttentionTransformer) and main.py:25-36 (VqVideoDiffusionModel)
as pure functions over a state_dict (key schema: SURVEY.md appendix A).
"""
import torch
import torch.nn.functional as F

from .attention import attention_module

LN_EPS = 1e-5  # nn.LayerNorm default, local_3d_attention.py:14


def depth_of(params, prefix='transformer.'):
    n = 0
    while f'{prefix}layers.{n}.0.norm.weight' in params:
        n += 1
    return n


def embed_tokens(params, z, prefix='transformer.'):
    """Token embedding + 3-axis position embedding (local_3d_attention.py:140-157).

    z: int64 [B,S,H,W]; S may be shorter than the table (reference test() :168-171)."""
    B, S, H, W = z.shape
    x = params[prefix + 'embedding.weight'][z]
    ps = params[prefix + 'pos_emb_s.weight'][:S].view(1, S, 1, 1, -1)
    ph = params[prefix + 'pos_emb_h.weight'][:H].view(1, 1, H, 1, -1)
    pw = params[prefix + 'pos_emb_w.weight'][:W].view(1, 1, 1, W, -1)
    # reference sums (s + h) + w then adds to x (:149-151, :157)
    return x + ((ps + ph) + pw)


def feed_forward(params, prefix, x):
    """Linear -> GELU(erf) -> Linear (local_3d_attention.py:20-31; dropout p=0)."""
    h = F.linear(x, params[prefix + 'net.0.weight'], params[prefix + 'net.0.bias'])
    h = F.gelu(h)
    return F.linear(h, params[prefix + 'net.3.weight'], params[prefix + 'net.3.bias'])


def layer_norm(params, prefix, x):
    w = params[prefix + 'norm.weight']
    return F.layer_norm(x, (x.shape[-1],), w, params[prefix + 'norm.bias'], LN_EPS)


def transformer_layer(params, lp, x, extents, heads):
    """One `x = attn(x, q=x) + x; x = ff(x) + x` step (local_3d_attention.py:159-161).

    Quirk Q1: PreNorm normalises only the positional argument, so k and v see
    LayerNorm(x) while q is projected from the raw residual stream (:16-17, :160)."""
    a = attention_module(params, lp + '0.fn.', layer_norm(params, lp + '0.', x), x, extents, heads)
    x = a + x
    f = feed_forward(params, lp + '1.fn.', layer_norm(params, lp + '1.', x))
    return f + x


def transformer_forward(params, z, extents, heads, prefix='transformer.', return_hidden=False):
    """Local3dAttentionTransformer.forward (local_3d_attention.py:153-163). No final LayerNorm."""
    x = embed_tokens(params, z, prefix)
    hidden = [x]
    for l in range(depth_of(params, prefix)):
        x = transformer_layer(params, f'{prefix}layers.{l}.', x, extents, heads)
        hidden.append(x)
    if return_hidden:
        return x, hidden
    return x


def denoiser_forward(params, z, extents, heads, batch_chunk=None):
    """VqVideoDiffusionModel.forward (main.py:33-36): logits of the LAST frame only.

    batch_chunk bounds memory at the BASELINE.json sizes (clips are independent)."""
    if batch_chunk is None or batch_chunk >= z.shape[0]:
        x = transformer_forward(params, z, extents, heads)
        return F.linear(x[:, -1], params['logit_proj.weight'], params['logit_proj.bias'])
    outs = [denoiser_forward(params, z[b:b + batch_chunk], extents, heads)
            for b in range(0, z.shape[0], batch_chunk)]
    return torch.cat(outs, dim=0)


# ---------------------------------------------------------------------------
# config 5: sparse dense-attention model (minecraft/sparse_diffusion.py:75-111,
# minecraft/transformer.py:34-80)
# ---------------------------------------------------------------------------

def dense_attention(params, prefix, x, heads):
    """lucidrains ViT attention: fused to_qkv (no bias), softmax(QK^T * scale) V, to_out
    (minecraft/transformer.py:34-63)."""
    B, n, _ = x.shape
    qkv = F.linear(x, params[prefix + 'to_qkv.weight'])
    q, k, v = qkv.chunk(3, dim=-1)
    dh = q.shape[-1] // heads

    def split(t):
        return t.reshape(B, n, heads, dh).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    dots = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5
    out = torch.matmul(torch.softmax(dots, dim=-1), v)
    out = out.transpose(1, 2).reshape(B, n, heads * dh)
    if prefix + 'to_out.0.weight' in params:
        out = F.linear(out, params[prefix + 'to_out.0.weight'], params[prefix + 'to_out.0.bias'])
    return out


def sparse_denoiser_forward(params, tokens, indices, shape, heads):
    """VqSparseDiffusionModel.forward(input, indices) (minecraft/sparse_diffusion.py:91-111).

    tokens, indices: int64 [B,n]; indices are flat positions in the (S,H,W) grid."""
    S, H, W = shape
    w_pos = indices % W
    h_pos = torch.div(indices, W, rounding_mode='trunc') % H
    s_pos = torch.div(indices, H * W, rounding_mode='trunc')
    x = params['embedding.weight'][tokens]
    x = x + (params['pos_emb_s.weight'][s_pos] + params['pos_emb_h.weight'][h_pos]
             + params['pos_emb_w.weight'][w_pos])
    l = 0
    while f'transformer.layers.{l}.0.norm.weight' in params:
        lp = f'transformer.layers.{l}.'
        x = dense_attention(params, lp + '0.fn.', layer_norm(params, lp + '0.', x), heads) + x
        x = feed_forward(params, lp + '1.fn.', layer_norm(params, lp + '1.', x)) + x
        l += 1
    return F.linear(x, params['logit_proj.weight'], params['logit_proj.bias'])
