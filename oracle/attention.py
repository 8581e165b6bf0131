"""Oracle (test infrastructure): local windowed 3D attention, fp32 CPU.

Restates vq-video-diffusion/local_3d_attention.py:57-99 (pad / unfold /
get_mask / local_attention) and :102-118 (Local3dAttention.forward).

The reference materialises a (N, heads, K, dh) copy of the unfolded keys and
values.  This restatement walks the K = (2eS+1)(2eH+1)(2eW+1) window offsets
instead: for offset o = (i, j, k) (row-major, the reference's `(i j k)` order)
the neighbour of token (s, h, w) is (s+i-eS, h+j-eH, w+k-eW); out-of-grid
neighbours contribute k = v = 0 and their logit is overwritten with -1e9
(reference :82-83, :92-94).  Same arithmetic, O(N*K) memory for the logits
only, so it also runs at the full BASELINE.json sizes (chunked over the batch).
"""
import torch
import torch.nn.functional as F

MASK_VALUE = -1e9  # local_3d_attention.py:92


def window_offsets(extents):
    """(i, j, k) offsets in the reference's unfold order (local_3d_attention.py:65-69, :86-87)."""
    eS, eH, eW = extents
    return [(i, j, k)
            for i in range(2 * eS + 1)
            for j in range(2 * eH + 1)
            for k in range(2 * eW + 1)]


def _pad3(x, extents, value=0.0):
    """Zero-pad the S, H, W axes of x[B,S,H,W,C] by the extents (local_3d_attention.py:57-63)."""
    eS, eH, eW = extents
    return F.pad(x, (0, 0, eW, eW, eH, eH, eS, eS), value=value)


def pad_mask(shape, extents):
    """bool[S,H,W,K]: True where window slot o of token (s,h,w) falls outside the grid
    (local_3d_attention.py:71-76)."""
    _, S, H, W = shape[:4]
    eS, eH, eW = extents
    inside = torch.zeros(1, S, H, W, 1)
    padded = _pad3(inside, extents, value=1.0)[0, ..., 0] > 0.5  # True on padding
    cols = [padded[i:i + S, j:j + H, k:k + W] for (i, j, k) in window_offsets(extents)]
    return torch.stack(cols, dim=-1)


def local_attention_logits(q, k, extents, heads):
    """Masked, scaled logits [B,S,H,W,heads,K] (local_3d_attention.py:89-94).

    q, k: [B,S,H,W,heads*dh] with the head index major inside the channel axis
    (the reference's `(H d)` split, :85-87)."""
    B, S, H, W, I = q.shape
    dh = I // heads
    scale = dh ** -0.5
    kp = _pad3(k, extents)
    qh = q.reshape(B, S, H, W, heads, dh)
    cols = []
    for (i, j, kk) in window_offsets(extents):
        ks = kp[:, i:i + S, j:j + H, kk:kk + W].reshape(B, S, H, W, heads, dh)
        cols.append((qh * ks).sum(-1))
    dots = torch.stack(cols, dim=-1) * scale            # [B,S,H,W,heads,K]
    m = pad_mask(q.shape, extents)                      # [S,H,W,K]
    dots = dots.masked_fill(m[None, :, :, :, None, :], MASK_VALUE)
    return dots


def local_attention(k, v, q, extents, heads, return_logits=False):
    """Attention core (local_3d_attention.py:78-99) followed by the
    'b h n d -> b n (h d)' merge of :113.  Returns [B,S,H,W,heads*dh]."""
    B, S, H, W, I = q.shape
    dh = I // heads
    dots = local_attention_logits(q, k, extents, heads)
    attn = torch.softmax(dots, dim=-1)                  # :96
    vp = _pad3(v, extents)
    out = torch.zeros(B, S, H, W, heads, dh, dtype=q.dtype)
    for o, (i, j, kk) in enumerate(window_offsets(extents)):
        vs = vp[:, i:i + S, j:j + H, kk:kk + W].reshape(B, S, H, W, heads, dh)
        out = out + attn[..., o:o + 1] * vs             # :97
    out = out.reshape(B, S, H, W, I)
    if return_logits:
        return out, dots
    return out


def local_attention_lse(q, k, extents, heads):
    """log-sum-exp of the masked logits per (token, head): what the HIP forward
    saves for its backward instead of the reference's checkpoint re-run (:110-111)."""
    return torch.logsumexp(local_attention_logits(q, k, extents, heads), dim=-1)


def attention_module(params, prefix, x, q, extents, heads):
    """Local3dAttention.forward(x, q) (local_3d_attention.py:102-118).

    params: state_dict-style mapping; keys prefix+'to_q.weight', 'to_k.weight',
    'to_v.weight', 'to_v.bias' and, unless heads==1 and dim_head==dim
    (:40, :50-53), 'to_out.0.weight' / 'to_out.0.bias'."""
    q_shape = q.shape
    kk = F.linear(x, params[prefix + 'to_k.weight'])                              # :106
    vv = F.linear(x, params[prefix + 'to_v.weight'], params[prefix + 'to_v.bias'])  # :107
    qq = F.linear(q, params[prefix + 'to_q.weight'])                              # :108
    out = local_attention(kk, vv, qq, extents, heads)
    if prefix + 'to_out.0.weight' in params:
        out = F.linear(out, params[prefix + 'to_out.0.weight'], params[prefix + 'to_out.0.bias'])
    return out.reshape(q_shape)
