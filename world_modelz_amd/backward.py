"""Backward passes of the fused blocks in functional.py, written over the HIP kernels (no autograd through
torch ops on the hot path).  Parameter gradients are fp32; activation gradients stay in the compute dtype."""
import torch

from . import _cast, ops
from .functional import LN_EPS


def _wt(w, dtype, tag):
    """W^T contiguous in the operand dtype (cached per parameter version): the B operand of dA = dC @ W."""
    return _cast.operand((w,), dtype, tag, lambda a: a.t())


def _flat(t):
    return t.reshape(-1, t.shape[-1])


def attention_block_backward(ctx, dy):
    x_kv, x_q, ln_g, ln_b, wq, wk, wv, bv, wout, bout, q, kv, o, lse = ctx.saved_tensors
    dt = x_kv.dtype
    I = wq.shape[0]
    dy = dy.contiguous()
    same_src = x_kv.data_ptr() == x_q.data_ptr() and x_kv.shape == x_q.shape
    grads = {}
    # ---- to_out (+ residual): y = o Wout^T + bout + residual
    d_res = dy if ctx.has_res else None
    if wout is not None:
        do = ops.linear_dgrad(dy, _wt(wout, dt, 'woutT'))
        grads['wout'] = torch.zeros_like(wout, dtype=torch.float32)
        grads['bout'] = torch.zeros_like(bout, dtype=torch.float32)
        ops.linear_wgrad(dy, o, grads['wout'], grads['bout'])
    else:
        do = dy
    # ---- attention core
    dq, dkv = ops.local3d_attention_bwd(q, kv[..., :I], kv[..., I:], o, lse, do, ctx.extents, ctx.heads)
    # ---- to_q on the raw input
    grads['wq'] = torch.zeros_like(wq, dtype=torch.float32)
    ops.linear_wgrad(dq, x_q, grads['wq'])
    dxq = ops.linear_dgrad(dq, _wt(wq, dt, 'wqT'))
    # ---- to_k | to_v on LN(x_kv)
    dwkv = torch.zeros((2 * I, wk.shape[1]), dtype=torch.float32, device=dy.device)
    dbkv = torch.zeros((2 * I,), dtype=torch.float32, device=dy.device)
    wkvT = _cast.operand((wk, wv), dt, 'kvT', lambda a, b: torch.cat([a, b], dim=0).t())
    dxhat = ops.linear_dgrad(dkv, wkvT)                      # gradient w.r.t. LN(x_kv) (or x_kv without a norm)
    fold_q = same_src                                        # attn(x, q=x): the q path lands on the same tensor
    fold_res = ctx.has_res and ctx.res_is_xkv                # ... + x: so does the residual path
    skip = None
    if fold_q:
        skip = dxq
    if fold_res:
        skip = d_res if skip is None else skip + d_res
    if ln_g is not None:
        stats = ops.layernorm_stats(x_kv, LN_EPS)
        ops.linear_wgrad(dkv, x_kv, dwkv, dbkv, ln=(ln_g.detach(), ln_b.detach()), ln_stats=stats)
        grads['ln_g'] = torch.zeros_like(ln_g, dtype=torch.float32)
        grads['ln_b'] = torch.zeros_like(ln_b, dtype=torch.float32)
        dx_kv = ops.layernorm_bwd(x_kv, dxhat, ln_g.detach(), grads['ln_g'], grads['ln_b'], skip=skip, eps=LN_EPS)
    else:
        ops.linear_wgrad(dkv, x_kv, dwkv, dbkv)
        dx_kv = dxhat if skip is None else dxhat + skip.reshape(dxhat.shape)
    grads['wk'], grads['wv'], grads['bv'] = dwkv[:I], dwkv[I:], dbkv[I:]
    g_xkv = dx_kv.reshape(x_kv.shape)
    g_xq = None if fold_q else dxq.reshape(x_q.shape)
    g_res = None if (not ctx.has_res or fold_res) else d_res
    return (g_xkv, g_xq, grads.get('ln_g'), grads.get('ln_b'), grads['wq'], grads['wk'], grads['wv'], grads['bv'],
            grads.get('wout'), grads.get('bout'), g_res, None, None, None)


def feed_forward_block_backward(ctx, dy):
    x, ln_g, ln_b, w1, b1, w2, b2, z = ctx.saved_tensors
    dt = x.dtype
    dy = dy.contiguous()
    d_res = dy if ctx.has_res else None
    # y = GELU(z) W2^T + b2 (+ residual)
    dw2 = torch.zeros_like(w2, dtype=torch.float32)
    db2 = torch.zeros_like(b2, dtype=torch.float32)
    ops.linear_wgrad(dy, z, dw2, db2, gelu_in=True)
    dz = ops.linear_dgrad(dy, _wt(w2, dt, 'w2T'), dgelu_z=z)        # (dy W2) * gelu'(z)
    # z = LN(x) W1^T + b1
    dw1 = torch.zeros_like(w1, dtype=torch.float32)
    db1 = torch.zeros_like(b1, dtype=torch.float32)
    dxhat = ops.linear_dgrad(dz, _wt(w1, dt, 'w1T'))
    dg = db = None
    if ln_g is not None:
        stats = ops.layernorm_stats(x, LN_EPS)
        ops.linear_wgrad(dz, x, dw1, db1, ln=(ln_g.detach(), ln_b.detach()), ln_stats=stats)
        dg = torch.zeros_like(ln_g, dtype=torch.float32)
        db = torch.zeros_like(ln_b, dtype=torch.float32)
        # the transformer passes residual = x: fold the skip gradient into the LayerNorm backward
        fold = ctx.has_res and ctx.res_is_x
        dx = ops.layernorm_bwd(x, dxhat, ln_g.detach(), dg, db, skip=d_res if fold else None, eps=LN_EPS)
        g_res = None if fold else d_res
    else:
        ops.linear_wgrad(dz, x, dw1, db1)
        dx = dxhat
        g_res = d_res
    return dx.reshape(x.shape), dg, db, dw1, db1, dw2, db2, g_res, None


def embed_backward(ctx, dx):
    (z,) = ctx.saved_tensors
    demb, dps, dph, dpw = ops.embed_pos3d_bwd(z, dx, ctx.shapes)
    return None, demb, dps, dph, dpw, None


def linear_backward(ctx, dy):
    x, w, b = ctx.saved_tensors
    dt = x.dtype
    dyc = dy.to(dt).contiguous()                                   # fused CE hands it over in dt already; torch's CE in fp32
    dw = torch.zeros_like(w, dtype=torch.float32)
    db = torch.zeros_like(b, dtype=torch.float32) if b is not None else None
    xc = x.contiguous()
    ops.linear_wgrad(dyc, xc, dw, db)
    dx = ops.linear_dgrad(dyc, _wt(w, dt, 'wT'))
    return dx.reshape(x.shape), dw, db, None


def dense_attention_block_backward(ctx, dy):
    from .functional import _dense_grid
    x, ln_g, ln_b, wqkv, wout, bout, qkv, o, lse = ctx.saved_tensors
    dt = x.dtype
    B, n, _ = x.shape
    I = wqkv.shape[0] // 3
    dy = dy.contiguous()
    d_res = dy if ctx.has_res else None
    dwout = dbout = None
    if wout is not None:
        do = ops.linear_dgrad(dy, _wt(wout, dt, 'woutT'))
        dwout = torch.zeros_like(wout, dtype=torch.float32)
        dbout = torch.zeros_like(bout, dtype=torch.float32)
        ops.linear_wgrad(dy, o, dwout, dbout)
    else:
        do = dy
    (S, H, W), ext = _dense_grid(n)
    g = qkv.view(B, S, H, W, 3 * I)
    dqkv = torch.empty_like(qkv)
    ops.local3d_attention_bwd(g[..., :I], g[..., I:2 * I], g[..., 2 * I:], o.view(B, S, H, W, I), lse,
                              do.view(B, S, H, W, I), ext, ctx.heads, dqkv=dqkv.view(B, S, H, W, 3 * I))
    dwqkv = torch.zeros_like(wqkv, dtype=torch.float32)
    dxhat = ops.linear_dgrad(dqkv, _wt(wqkv, dt, 'wqkvT'))
    dg = db = None
    fold = ctx.has_res and ctx.res_is_x
    if ln_g is not None:
        stats = ops.layernorm_stats(x, LN_EPS)
        ops.linear_wgrad(dqkv, x, dwqkv, None, ln=(ln_g.detach(), ln_b.detach()), ln_stats=stats)
        dg = torch.zeros_like(ln_g, dtype=torch.float32)
        db = torch.zeros_like(ln_b, dtype=torch.float32)
        dx = ops.layernorm_bwd(x, dxhat, ln_g.detach(), dg, db, skip=d_res if fold else None, eps=LN_EPS)
    else:
        ops.linear_wgrad(dqkv, x, dwqkv, None)
        dx = dxhat + d_res if fold else dxhat
    g_res = None if (not ctx.has_res or fold) else d_res
    return dx.reshape(x.shape), dg, db, dwqkv, dwout, dbout, g_res, None, None
