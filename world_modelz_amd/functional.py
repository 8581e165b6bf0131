"""Fused building blocks of the denoiser as torch.autograd.Functions over the HIP kernels.

attention_block : to_out(local_attention(to_q(x_q), to_k(LN?(x_kv)), to_v(LN?(x_kv)))) [+ residual]
                  (reference: PreNorm + Local3dAttention + the `+ x` of the layer loop,
                  local_3d_attention.py:16-17, :102-118, :160)
feed_forward_block : W2 GELU(W1 LN?(x) + b1) + b2 [+ residual]     (local_3d_attention.py:20-31, :161)
embed_tokens    : token + 3-axis position embedding                 (local_3d_attention.py:140-157)

Parameters arrive as fp32 nn.Parameters; MFMA-operand copies in the compute dtype are cached per parameter
version (_cast.operand).  Gradients of parameters are returned in fp32.
"""
import torch

from . import _cast, ops
from .config import get_compute_dtype

LN_EPS = 1e-5


class _AttentionBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_kv, x_q, ln_g, ln_b, wq, wk, wv, bv, wout, bout, residual, extents, heads, grad_on, same_src,
                res_is_xkv, out_f32=False):
        dt = x_kv.dtype
        I = wq.shape[0]
        ln = None if ln_g is None else (ln_g.detach(), ln_b.detach())
        wq_c = _cast.operand(wq, dt)
        wkv_c = _cast.operand((wk, wv), dt, 'kv', lambda a, b: torch.cat([a, b], dim=0))
        bkv = _cast.operand((bv,), torch.float32, 'bkv', lambda b: torch.cat([torch.zeros_like(b), b]))
        lead = x_q.shape[:-1]
        need_bwd = grad_on and any(ctx.needs_input_grad)     # grad mode is always off inside forward()
        # training: the LayerNorm statistics are computed once, handed to the GEMM prologue and kept for the backward
        stats = ops.layernorm_stats(x_kv, LN_EPS) if (need_bwd and ln is not None) else None
        q = ops.linear_fwd(x_q, wq_c)                                           # to_q: no bias, raw input (Q1)
        xn = None
        if stats is not None and x_kv.numel() // x_kv.shape[-1] >= ops.KEEP_NORM_MIN_ROWS:
            kv, _, xn = ops.linear_fwd_train(x_kv, wkv_c, bkv, ln, LN_EPS, stats, want_norm=True)     # (+ LN(x) for the weight gradient)
        else:
            kv = ops.linear_fwd(x_kv, wkv_c, bias=bkv, ln=ln, ln_eps=LN_EPS, ln_stats=stats)   # to_k | to_v on LN(x)
        o, lse, _ = ops.local3d_attention_fwd(q, kv[..., :I], kv[..., I:], extents, heads, need_lse=need_bwd)
        if wout is not None:
            # (inference at a module boundary that wants fp32 back -- Local3dAttention.forward on fp32 inputs: the to_out GEMM's
            #  epilogue writes it, instead of a cast launch behind the block)
            y = ops.linear_fwd(o, _cast.operand(wout, dt), bias=bout.detach(), residual=residual, out_f32=bool(out_f32) and not need_bwd)
        else:
            y = o if residual is None else o + residual
        if need_bwd:
            ctx.save_for_backward(x_kv, x_q, ln_g, ln_b, wq, wk, wv, bv, wout, bout, q, kv, o, lse)
            ctx.extents, ctx.heads, ctx.has_res = extents, heads, residual is not None
            ctx.ln_stats = stats
            ctx.xn = xn                    # LN(x_kv) as the k | v GEMM consumed it (None: the weight gradient re-normalises)
            # gradient folding is decided by autograd identity of the caller's tensors (attention_block), never by
            # storage: attn(x, q=x.detach()) or an aliased residual must get separate gradients
            ctx.same_src, ctx.res_is_xkv = bool(same_src), bool(res_is_xkv)
        return y.reshape(*lead, y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        from . import backward as B
        return B.attention_block_backward(ctx, dy) + (None,)


class _FeedForwardBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_g, ln_b, w1, b1, w2, b2, residual, grad_on, res_is_x):
        dt = x.dtype
        ln = None if ln_g is None else (ln_g.detach(), ln_b.detach())
        need_bwd = grad_on and any(ctx.needs_input_grad)
        w1_c, w2_c = _cast.operand(w1, dt), _cast.operand(w2, dt)
        if need_bwd:
            # keep the pre-activation z (gelu' in the backward) AND the activation h = GELU(z) (the second GEMM's operand here,
            # its weight gradient's operand in the backward): one launch writes both
            stats = ops.layernorm_stats(x, LN_EPS) if ln is not None else None
            xn = None
            if ln is not None and x.numel() // x.shape[-1] >= ops.KEEP_NORM_MIN_ROWS:
                # many rows: LN(x) is kept as well -- the first GEMM's weight gradient then is a plain GEMM (256-wide tiles)
                z, h, xn = ops.linear_fwd_train(x, w1_c, b1.detach(), ln, LN_EPS, stats, want_gelu=True, want_norm=True)
            else:
                z, h = ops.linear_fwd_gelu_pair(x, w1_c, bias=b1.detach(), ln=ln, ln_eps=LN_EPS, ln_stats=stats)
            y = ops.linear_fwd(h, w2_c, bias=b2.detach(), residual=residual)
            ctx.save_for_backward(x, ln_g, ln_b, w1, b1, w2, b2, z, h, xn)
            ctx.ln_stats = stats
            ctx.has_res = residual is not None
            ctx.res_is_x = bool(res_is_x)
        else:
            h = ops.linear_fwd(x, w1_c, bias=b1.detach(), ln=ln, ln_eps=LN_EPS, gelu=True)
            y = ops.linear_fwd(h, w2_c, bias=b2.detach(), residual=residual)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import backward as B
        return B.feed_forward_block_backward(ctx, dy)


class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, emb, pos_s, pos_h, pos_w, dtype):
        x = ops.embed_pos3d_fwd(z, emb.detach(), pos_s.detach(), pos_h.detach(), pos_w.detach(), dtype)
        ctx.save_for_backward(z)
        ctx.params = (emb, pos_s, pos_h, pos_w)
        return x

    @staticmethod
    def backward(ctx, dx):
        from . import backward as B
        return B.embed_backward(ctx, dx)


class _Linear(torch.autograd.Function):
    """Plain nn.Linear on the HIP GEMM (logit_proj, main.py:31-36); fp32 output optional."""

    @staticmethod
    def forward(ctx, x, w, b, out_f32):
        bias = None if b is None else b.detach()
        if x.dim() >= 3 and not x.is_contiguous() and x[0].is_contiguous() and x.stride(0) % 8 == 0:
            # a batch of contiguous row blocks a fixed stride apart -- the last frame x[:, -1] of the stream: read in place
            xb = x.view(x.shape[0], -1, x.shape[-1])
            y = ops.linear_fwd_blocks(xb, _cast.operand(w, x.dtype), bias, out_f32).view(*x.shape[:-1], w.shape[0])
        else:
            y = ops.linear_fwd(x, _cast.operand(w, x.dtype), bias=bias, out_f32=out_f32)
        ctx.save_for_backward(x, w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import backward as B
        return B.linear_backward(ctx, dy)


def _as_compute(x):
    dt = get_compute_dtype()
    if not x.is_cuda:
        raise ops.L.WmzError('world_modelz_amd modules run on the GPU only (no CPU fallback); got a CPU tensor')
    return x if x.dtype == dt else x.to(dt)


def attention_block(x_kv, x_q, ln, wq, wk, wv, bv, wout, bout, residual, extents, heads, out_f32=False):
    g, b = (None, None) if ln is None else ln
    return _AttentionBlock.apply(x_kv, x_q, g, b, wq, wk, wv, bv, wout, bout, residual, tuple(int(e) for e in extents),
                                 int(heads), torch.is_grad_enabled(), x_q is x_kv, residual is x_kv, bool(out_f32))


def feed_forward_block(x, ln, w1, b1, w2, b2, residual):
    g, b = (None, None) if ln is None else ln
    return _FeedForwardBlock.apply(x, g, b, w1, b1, w2, b2, residual, torch.is_grad_enabled(), residual is x)


def embed_tokens(z, emb, pos_s, pos_h, pos_w):
    return _Embed.apply(z, emb, pos_s, pos_h, pos_w, get_compute_dtype())


def linear(x, w, b=None, out_f32=False):
    return _Linear.apply(x, w, b, out_f32)


# ------------------------------------------------------------------------------------------------ config 5 (dense ViT)

def _dense_grid(n):
    """A length-n sequence as a token grid whose window covers everything: dense attention IS local attention with
    unbounded extents, so the same kernels serve it (16-wide planes take the fast path)."""
    if n % 16 == 0:
        return (1, n // 16, 16), (0, n // 16, 16)
    return (1, 1, n), (0, 0, n)


class _DenseAttentionBlock(torch.autograd.Function):
    """to_out(softmax(q k^T * scale) v) + residual with q|k|v = to_qkv(LN(x)) (no bias): PreNorm(Attention) of
    minecraft/transformer.py:11-63 plus the `+ x` of :77."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wqkv, wout, bout, residual, heads, grad_on, res_is_x):
        dt = x.dtype
        B, n, _ = x.shape
        I = wqkv.shape[0] // 3
        ln = None if ln_g is None else (ln_g.detach(), ln_b.detach())
        need_bwd = grad_on and any(ctx.needs_input_grad)
        # training: the LayerNorm statistics are computed ONCE (the GEMM's twelve column tiles would each redo the two passes
        # over their rows) and kept for the backward
        stats = ops.layernorm_stats(x, LN_EPS) if (ln is not None and need_bwd) else None
        qkv = ops.linear_fwd(x, _cast.operand(wqkv, dt), ln=ln, ln_eps=LN_EPS, ln_stats=stats)    # [B, n, 3I]
        (S, H, W), ext = _dense_grid(n)
        g = qkv.view(B, S, H, W, 3 * I)
        o, lse, _ = ops.local3d_attention_fwd(g[..., :I], g[..., I:2 * I], g[..., 2 * I:], ext, heads, need_lse=need_bwd)
        o = o.view(B, n, I)
        if wout is not None:
            y = ops.linear_fwd(o, _cast.operand(wout, dt), bias=bout.detach(), residual=residual)
        else:
            y = o if residual is None else o + residual
        if need_bwd:
            ctx.save_for_backward(x, ln_g, ln_b, wqkv, wout, bout, qkv, o, lse)
            ctx.ln_stats = stats
            ctx.heads, ctx.has_res = heads, residual is not None
            ctx.res_is_x = bool(res_is_x)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import backward as Bk
        return Bk.dense_attention_block_backward(ctx, dy)


class _EmbedIndexed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, pos, emb, pos_s, pos_h, pos_w, shape, dtype):
        x = ops.embed_indexed_fwd(tok, pos, emb.detach(), pos_s.detach(), pos_h.detach(), pos_w.detach(), shape, dtype)
        ctx.save_for_backward(tok, pos)
        ctx.shape3, ctx.shapes = shape, (emb.shape, pos_s.shape, pos_h.shape, pos_w.shape)
        ctx.params = (emb, pos_s, pos_h, pos_w)
        return x

    @staticmethod
    def backward(ctx, dx):
        tok, pos = ctx.saved_tensors
        bufs = [getattr(p, '_wmz_grad', None) for p in ctx.params]
        if all(b is not None for b in bufs):
            # flat gradient arena: the kernel's atomics land in the parameters' slices (as embed_backward does for the grid model)
            ops.embed_indexed_bwd(tok, pos, dx, ctx.shape3, ctx.shapes, into=bufs)
            for p in ctx.params:
                ready = getattr(p, '_wmz_ready', None)
                if ready is not None:
                    ready()
            return None, None, None, None, None, None, None, None
        demb, dps, dph, dpw = ops.embed_indexed_bwd(tok, pos, dx, ctx.shape3, ctx.shapes)
        return None, None, demb, dps, dph, dpw, None, None


def dense_attention_block(x, ln, wqkv, wout, bout, residual, heads):
    g, b = (None, None) if ln is None else ln
    return _DenseAttentionBlock.apply(x, g, b, wqkv, wout, bout, residual, int(heads), torch.is_grad_enabled(),
                                      residual is x)


def embed_tokens_indexed(tok, pos, emb, pos_s, pos_h, pos_w, shape):
    return _EmbedIndexed.apply(tok, pos, emb, pos_s, pos_h, pos_w, tuple(int(s) for s in shape), get_compute_dtype())
