"""Tensor-level wrappers over the C ABI (shape/stride checks, output allocation).  No autograd here."""
import torch

from . import _lib as L

_profile_hook = None


def set_profile_hook(fn):
    """fn(kernel_name, is_start) is called right before / after a launch is enqueued (bench.py uses it to
    bracket one kernel with HIP events on the launch stream)."""
    global _profile_hook
    _profile_hook = fn


def _rows(t):
    """View [..., C] as rows; returns (tensor, nrows, row stride in elements). Last dim must be contiguous."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    C = t.shape[-1]
    flat = t.reshape(-1, C) if t.is_contiguous() else None
    if flat is None:
        # allow a column slice of a contiguous [N, C_total] buffer (fused qkv): uniform row stride
        if t.dim() >= 2 and all(t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2)):
            return t, t.numel() // C, t.stride(-2)
        t = t.contiguous()
        flat = t.reshape(-1, C)
    return t, flat.shape[0], C


def local3d_attention_fwd(q, k, v, extents, heads, need_lse=False, logits_dbg=False):
    """q, k, v: [B,S,H,W,heads*dh] (last dim contiguous, uniform row stride).  Returns (out, lse, logits)."""
    B, S, H, W, I = q.shape
    dh = I // heads
    dt = L.dtype_code(q.dtype)
    q, _, ldq = _rows(q)
    k, _, ldk = _rows(k)
    v, _, ldv = _rows(v)
    out = torch.empty((B, S, H, W, I), dtype=q.dtype, device=q.device)
    N = B * S * H * W
    lse = torch.empty((N, heads), dtype=torch.float32, device=q.device) if need_lse else None
    dbg = None
    if logits_dbg:
        K = (2 * extents[0] + 1) * (2 * extents[1] + 1) * (2 * extents[2] + 1)
        dbg = torch.full((N, heads, K), -1e9, dtype=torch.float32, device=q.device)
    if _profile_hook is not None:
        _profile_hook('wmz_local3d_attn_fwd', True)
    L.call('wmz_local3d_attn_fwd', L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(dbg),
           B, S, H, W, heads, dh, int(extents[0]), int(extents[1]), int(extents[2]), ldq, ldk, ldv, I, dt, L.stream())
    if _profile_hook is not None:
        _profile_hook('wmz_local3d_attn_fwd', False)
    return out, lse, dbg


def linear_fwd(a, weight, bias=None, residual=None, ln=None, ln_eps=1e-5, gelu=False, gelu_in=False, out_f32=False,
               out=None):
    """act(LN?(a) @ weight^T + bias) + residual.   a: [..., K]; weight: [N, K] in a's dtype; bias / ln fp32."""
    K = a.shape[-1]
    N = weight.shape[0]
    dt = L.dtype_code(a.dtype)
    assert weight.dtype == a.dtype and weight.is_contiguous() and weight.shape[1] == K
    a, M, lda = _rows(a)
    lead = a.shape[:-1]
    if out is None:
        out = torch.empty(lead + (N,), dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    out_t, Mo, ldc = _rows(out)
    assert Mo == M and out_t.data_ptr() == out.data_ptr()
    ldr = 0
    if residual is not None:
        residual, Mr, ldr = _rows(residual)
        assert Mr == M and residual.dtype == a.dtype
    g = b = None
    if ln is not None:
        g, b = ln
        assert g.dtype == torch.float32 and b.dtype == torch.float32
    if bias is not None:
        assert bias.dtype == torch.float32
    L.call('wmz_linear_fwd', L.ptr(a), lda, L.ptr(weight), L.ptr(bias), L.ptr(residual), ldr, L.ptr(out), ldc,
           M, N, K, L.ptr(g), L.ptr(b), float(ln_eps),
           (L.WMZ_LIN_GELU if gelu else 0) | (L.WMZ_LIN_GELU_IN if gelu_in else 0), 1 if out_f32 else 0, dt,
           L.stream())
    return out


def embed_pos3d_fwd(z, emb, pos_s, pos_h, pos_w, dtype):
    B, S, H, W = z.shape
    D = emb.shape[1]
    z = z.contiguous()
    x = torch.empty((B, S, H, W, D), dtype=dtype, device=z.device)
    L.call('wmz_embed_pos3d_fwd', L.ptr(z), L.ptr(emb), L.ptr(pos_s), L.ptr(pos_h), L.ptr(pos_w), L.ptr(x),
           B, S, H, W, D, emb.shape[0], L.dtype_code(dtype), L.stream())
    return x


def vq_argmin(x, codebook, need_dist=False):
    """x: [N,E] fp32, codebook: [C,E] fp32 -> int64 [N] (+ fp32 min distance)."""
    assert x.dtype == torch.float32 and codebook.dtype == torch.float32
    x, N, ldx = _rows(x)
    codebook = codebook.contiguous()
    C, E = codebook.shape
    idx = torch.empty((N,), dtype=torch.int64, device=x.device)
    dmin = torch.empty((N,), dtype=torch.float32, device=x.device) if need_dist else None
    L.call('wmz_vq_argmin', L.ptr(x), ldx, L.ptr(codebook), L.ptr(idx), L.ptr(dmin), N, C, E, L.stream())
    return (idx, dmin) if need_dist else idx


def vq_gather(idx, codebook, dtype=torch.float32):
    codebook = codebook.contiguous()
    C, E = codebook.shape
    flat = idx.reshape(-1).contiguous()
    out = torch.empty((flat.shape[0], E), dtype=dtype, device=idx.device)
    L.call('wmz_vq_gather', L.ptr(flat), L.ptr(codebook), L.ptr(out), E, flat.shape[0], C, E, L.dtype_code(dtype),
           L.stream())
    return out.reshape(*idx.shape, E)


def vq_ema_stats(x, idx, codebook, counts=None, dw=None, sqerr=None):
    x, N, ldx = _rows(x)
    C, E = codebook.shape
    L.call('wmz_vq_ema_stats', L.ptr(x), ldx, L.ptr(idx.reshape(-1).contiguous()), L.ptr(codebook), L.ptr(counts),
           L.ptr(dw), L.ptr(sqerr), N, C, E, L.stream())


def vq_ema_update(embedding, cluster_size, activation_count, counts, dw, decay, eps):
    C, E = embedding.shape[-2:]
    L.call('wmz_vq_ema_update', L.ptr(embedding), L.ptr(cluster_size), L.ptr(activation_count), L.ptr(counts),
           L.ptr(dw), C, E, float(decay), float(eps), L.stream())
