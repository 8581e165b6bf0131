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


def _emit(param, fill):
    """Produce the gradient of `param` with `fill(buffer)` (every wgrad-type kernel ACCUMULATES into its output).

    Under a FlatArena (parallel.py) the parameter carries `_wmz_grad`, its slice of the flat gradient arena: the kernel
    accumulates straight into it, the data-parallel reducer is told (`_wmz_ready`) and autograd gets None -- no zeros
    allocation, no `grad += g` pass, no extra launches.  Otherwise a fresh fp32 tensor is returned for autograd."""
    if param is None:
        return None
    buf = getattr(param, '_wmz_grad', None)
    if buf is not None:
        with ops.arena_fill():              # (weight gradients may go to the side stream: ops.linear_wgrad)
            fill(buf)
        ready = getattr(param, '_wmz_ready', None)
        if ready is not None:
            ready()
        return None
    t = torch.zeros(param.shape, dtype=torch.float32, device=param.device)
    fill(t)
    return t


def _emit2(p1, p2, fill):
    """Two parameters filled by ONE kernel call (weight + bias, gamma + beta)."""
    b1 = getattr(p1, '_wmz_grad', None)
    b2 = getattr(p2, '_wmz_grad', None) if p2 is not None else None
    direct = b1 is not None and (p2 is None or b2 is not None)
    if not direct:
        b1 = torch.zeros(p1.shape, dtype=torch.float32, device=p1.device)
        b2 = torch.zeros(p2.shape, dtype=torch.float32, device=p2.device) if p2 is not None else None
    if direct:
        with ops.arena_fill():
            fill(b1, b2)
    else:
        fill(b1, b2)
    if direct:
        for p in (p1, p2):
            ready = getattr(p, '_wmz_ready', None) if p is not None else None
            if ready is not None:
                ready()
        return None, None
    return b1, b2


def _wgrad_kv(wk, wv, bv, dkv, x_kv, lnp, stats):
    """to_k and to_v weight gradients by ONE GEMM on dk | dv (they share the operand LN(x) and its prologue):
    dW[2I, D] = [dk | dv]^T LN(x).  Under a FlatArena the two gradient slices are adjacent (to_k.weight, to_v.weight are
    registered back to back), so the kernel accumulates straight into them.  Returns (g_wk, g_wv, g_bv) for autograd."""
    I = wk.shape[0]
    bk_, bw_, bb_ = (getattr(p, '_wmz_grad', None) for p in (wk, wv, bv))
    direct = (bk_ is not None and bw_ is not None and bb_ is not None and bk_.is_contiguous() and bw_.is_contiguous()
              and bw_.data_ptr() == bk_.data_ptr() + 4 * bk_.numel())
    db = torch.zeros(2 * I, dtype=torch.float32, device=dkv.device)
    if direct:
        dw = torch.as_strided(bk_, (2 * I, wk.shape[1]), (wk.shape[1], 1))
    else:
        dw = torch.zeros((2 * I, wk.shape[1]), dtype=torch.float32, device=dkv.device)
    if direct:
        with ops.arena_fill():          # (db is a scratch row: its to_v half joins the arena behind the launch, on the same stream)
            ops.linear_wgrad(dkv, x_kv, dw, db, ln=lnp, ln_stats=stats, then=lambda: bb_.add_(db[I:]))
    else:
        ops.linear_wgrad(dkv, x_kv, dw, db, ln=lnp, ln_stats=stats)
    if direct:
        for p in (wk, wv, bv):
            ready = getattr(p, '_wmz_ready', None)
            if ready is not None:
                ready()
        return None, None, None
    outs = []
    for p, g in ((wk, dw[:I]), (wv, dw[I:]), (bv, db[I:])):
        buf = getattr(p, '_wmz_grad', None)
        if buf is not None:
            buf += g
            ready = getattr(p, '_wmz_ready', None)
            if ready is not None:
                ready()
            outs.append(None)
        else:
            outs.append(g)
    return tuple(outs)


def attention_block_backward(ctx, dy):
    with ops.side_blocked(ctx.has_res and not ctx.res_is_xkv):      # dy itself goes back to autograd as the residual gradient
        return _attention_block_backward(ctx, dy)


def _attention_block_backward(ctx, dy):
    x_kv, x_q, ln_g, ln_b, wq, wk, wv, bv, wout, bout, q, kv, o, lse = ctx.saved_tensors
    dt = x_kv.dtype
    I = wq.shape[0]
    dy = dy.contiguous()
    same_src = ctx.same_src              # the caller passed ONE tensor as x and q (decided by identity in forward)
    # ---- to_out (+ residual): y = o Wout^T + bout + residual
    d_res = dy if ctx.has_res else None
    g_wout = g_bout = None
    if wout is not None:
        do = ops.linear_dgrad(dy, _wt(wout, dt, 'woutT'))
        g_wout, g_bout = _emit2(wout, bout, lambda w, b: ops.linear_wgrad(dy, o, w, b))
    else:
        do = dy
    # ---- attention core
    dq, dkv = ops.local3d_attention_bwd(q, kv[..., :I], kv[..., I:], o, lse, do, ctx.extents, ctx.heads)
    # ---- to_q on the raw input
    g_wq = _emit(wq, lambda w: ops.linear_wgrad(dq, x_q, w))
    dxq = ops.linear_dgrad(dq, _wt(wq, dt, 'wqT'))
    # ---- to_k | to_v on LN(x_kv)
    wkvT = _cast.operand((wk, wv), dt, 'kvT', lambda a, b: torch.cat([a, b], dim=0).t())
    dxhat = ops.linear_dgrad(dkv, wkvT)                      # gradient w.r.t. LN(x_kv) (or x_kv without a norm)
    fold_q = same_src                                        # attn(x, q=x): the q path lands on the same tensor
    fold_res = ctx.has_res and ctx.res_is_xkv                # ... + x: so does the residual path
    skip = dxq if fold_q else None
    skip2 = d_res if fold_res else None                      # both skips go into the LayerNorm backward: no separate add
    g_ln_g = g_ln_b = None
    dk, dv = dkv[..., :I], dkv[..., I:]
    if ln_g is not None:
        stats = ctx.ln_stats if getattr(ctx, 'ln_stats', None) is not None else ops.layernorm_stats(x_kv, LN_EPS)
        lnp = (ln_g.detach(), ln_b.detach())
        xn = getattr(ctx, 'xn', None)
        if xn is not None:
            g_wk, g_wv, g_bv = _wgrad_kv(wk, wv, bv, dkv, xn, None, None)         # plain operand: LN(x_kv) kept by the forward
        else:
            g_wk, g_wv, g_bv = _wgrad_kv(wk, wv, bv, dkv, x_kv, lnp, stats)
        holder = {}

        def fill_ln(gg, gb):
            holder['dx'] = ops.layernorm_bwd(x_kv, dxhat, ln_g.detach(), gg, gb, skip=skip, eps=LN_EPS, skip2=skip2)
        g_ln_g, g_ln_b = _emit2(ln_g, ln_b, fill_ln)
        dx_kv = holder['dx']
    else:
        g_wk, g_wv, g_bv = _wgrad_kv(wk, wv, bv, dkv, x_kv, None, None)
        dx_kv = dxhat
        for sk in (skip, skip2):
            if sk is not None:
                dx_kv = dx_kv + sk.reshape(dxhat.shape)
    g_xkv = dx_kv.reshape(x_kv.shape)
    g_xq = None if fold_q else dxq.reshape(x_q.shape)
    g_res = None if (not ctx.has_res or fold_res) else d_res
    return (g_xkv, g_xq, g_ln_g, g_ln_b, g_wq, g_wk, g_wv, g_bv, g_wout, g_bout, g_res, None, None, None, None, None)


def feed_forward_block_backward(ctx, dy):
    fold = ctx.has_res and ctx.res_is_x and ctx.saved_tensors[1] is not None
    with ops.side_blocked(ctx.has_res and not fold):
        return _feed_forward_block_backward(ctx, dy)


def _feed_forward_block_backward(ctx, dy):
    x, ln_g, ln_b, w1, b1, w2, b2, z = ctx.saved_tensors[:8]
    h = ctx.saved_tensors[8] if len(ctx.saved_tensors) > 8 else None      # GELU(z) as the forward stored it (or recompute in the loader)
    xn = ctx.saved_tensors[9] if len(ctx.saved_tensors) > 9 else None     # LN(x) as the first GEMM consumed it (many rows only)
    dt = x.dtype
    dy = dy.contiguous()
    d_res = dy if ctx.has_res else None
    # y = GELU(z) W2^T + b2 (+ residual)
    if h is not None:
        dw2, db2 = _emit2(w2, b2, lambda w, b: ops.linear_wgrad(dy, h, w, b))
    else:
        dw2, db2 = _emit2(w2, b2, lambda w, b: ops.linear_wgrad(dy, z, w, b, gelu_in=True))
    dz = ops.linear_dgrad(dy, _wt(w2, dt, 'w2T'), dgelu_z=z)        # (dy W2) * gelu'(z)
    # z = LN(x) W1^T + b1
    dxhat = ops.linear_dgrad(dz, _wt(w1, dt, 'w1T'))
    dg = db = None
    if ln_g is not None:
        stats = ctx.ln_stats if getattr(ctx, 'ln_stats', None) is not None else ops.layernorm_stats(x, LN_EPS)
        lnp = (ln_g.detach(), ln_b.detach())
        if xn is not None:
            dw1, db1 = _emit2(w1, b1, lambda w, b: ops.linear_wgrad(dz, xn, w, b))
        else:
            dw1, db1 = _emit2(w1, b1, lambda w, b: ops.linear_wgrad(dz, x, w, b, ln=lnp, ln_stats=stats))
        # the transformer passes residual = x: fold the skip gradient into the LayerNorm backward
        fold = ctx.has_res and ctx.res_is_x
        holder = {}

        def fill_ln(gg, gb):
            holder['dx'] = ops.layernorm_bwd(x, dxhat, ln_g.detach(), gg, gb, skip=d_res if fold else None, eps=LN_EPS)
        dg, db = _emit2(ln_g, ln_b, fill_ln)
        dx = holder['dx']
        g_res = None if fold else d_res
    else:
        dw1, db1 = _emit2(w1, b1, lambda w, b: ops.linear_wgrad(dz, x, w, b))
        dx = dxhat
        g_res = d_res
    return dx.reshape(x.shape), dg, db, dw1, db1, dw2, db2, g_res, None, None


def embed_backward(ctx, dx):
    (z,) = ctx.saved_tensors
    params = ctx.params                       # (emb, pos_s, pos_h, pos_w) Parameters
    bufs = [getattr(p, '_wmz_grad', None) for p in params]
    direct = all(b is not None for b in bufs)
    if not direct:
        bufs = [torch.zeros(p.shape, dtype=torch.float32, device=dx.device) for p in params]
    ops.embed_pos3d_bwd(z, dx, bufs)
    if direct:
        for p in params:
            ready = getattr(p, '_wmz_ready', None)
            if ready is not None:
                ready()
        return None, None, None, None, None, None
    return None, bufs[0], bufs[1], bufs[2], bufs[3], None


def linear_backward(ctx, dy):
    x, w, b = ctx.saved_tensors
    dt = x.dtype
    N = w.shape[0]
    if N % 8:
        # an output width that is no multiple of 8 (a vocabulary the reference accepts, main.py:404 --num_embeddings): the
        # gradient GEMMs run on zero-padded columns / weight rows, the padding's gradient rows are dropped
        import torch.nn.functional as F
        pad = -N % 8
        dyc = F.pad(dy.to(dt), (0, pad)).contiguous()
        xc = x.contiguous()
        gw = torch.zeros((N + pad, w.shape[1]), dtype=torch.float32, device=x.device)
        gb = torch.zeros(N + pad, dtype=torch.float32, device=x.device) if b is not None else None
        ops.linear_wgrad(dyc, xc, gw, gb)
        dx = ops.linear_dgrad(dyc, _cast.operand((w,), dt, 'wTpad8', lambda a: F.pad(a, (0, 0, 0, pad)).t()))
        return dx.reshape(x.shape), gw[:N], (gb[:N] if gb is not None else None), None
    dyc = dy.to(dt).contiguous()                                   # fused CE hands it over in dt already; torch's CE in fp32
    xc = x.contiguous()
    if b is not None:
        dw, db = _emit2(w, b, lambda gw, gb: ops.linear_wgrad(dyc, xc, gw, gb))
    else:
        dw, db = _emit(w, lambda gw: ops.linear_wgrad(dyc, xc, gw)), None
    dx = ops.linear_dgrad(dyc, _wt(w, dt, 'wT'))
    return dx.reshape(x.shape), dw, db, None


def dense_attention_block_backward(ctx, dy):
    with ops.side_blocked(ctx.has_res and not ctx.res_is_x):
        return _dense_attention_block_backward(ctx, dy)


def _dense_attention_block_backward(ctx, dy):
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
        dwout, dbout = _emit2(wout, bout, lambda w, b: ops.linear_wgrad(dy, o, w, b))
    else:
        do = dy
    (S, H, W), ext = _dense_grid(n)
    g = qkv.view(B, S, H, W, 3 * I)
    dqkv = torch.empty_like(qkv)
    ops.local3d_attention_bwd(g[..., :I], g[..., I:2 * I], g[..., 2 * I:], o.view(B, S, H, W, I), lse,
                              do.view(B, S, H, W, I), ext, ctx.heads, dqkv=dqkv.view(B, S, H, W, 3 * I))
    dxhat = ops.linear_dgrad(dqkv, _wt(wqkv, dt, 'wqkvT'))
    dg = db = None
    fold = ctx.has_res and ctx.res_is_x
    if ln_g is not None:
        stats = ctx.ln_stats if getattr(ctx, 'ln_stats', None) is not None else ops.layernorm_stats(x, LN_EPS)
        lnp = (ln_g.detach(), ln_b.detach())
        dwqkv = _emit(wqkv, lambda w: ops.linear_wgrad(dqkv, x, w, None, ln=lnp, ln_stats=stats))
        holder = {}

        def fill_ln(gg, gb):
            holder['dx'] = ops.layernorm_bwd(x, dxhat, ln_g.detach(), gg, gb, skip=d_res if fold else None, eps=LN_EPS)
        dg, db = _emit2(ln_g, ln_b, fill_ln)
        dx = holder['dx']
    else:
        dwqkv = _emit(wqkv, lambda w: ops.linear_wgrad(dqkv, x, w, None))
        dx = dxhat + d_res if fold else dxhat
    g_res = None if (not ctx.has_res or fold) else d_res
    return dx.reshape(x.shape), dg, db, dwqkv, dwout, dbout, g_res, None, None, None
