"""Tensor-level wrappers over the C ABI (shape/stride checks, output allocation).  No autograd here."""
import contextlib
import ctypes
import os

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


def local3d_attention_fwd(q, k, v, extents, heads, need_lse=False, logits_dbg=False, general=False):
    """q, k, v: [B,S,H,W,heads*dh] (last dim contiguous, uniform row stride).  Returns (out, lse, logits).
    general: take the general kernel even where the 16-wide-plane fast path applies (parity tests)."""
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
    L.call('wmz_local3d_attn_fwd_general' if general else 'wmz_local3d_attn_fwd', L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out),
           L.ptr(lse), L.ptr(dbg),
           B, S, H, W, heads, dh, int(extents[0]), int(extents[1]), int(extents[2]), ldq, ldk, ldv, I, dt, L.stream())
    if _profile_hook is not None:
        _profile_hook('wmz_local3d_attn_fwd', False)
    return out, lse, dbg


def linear_fwd(a, weight, bias=None, residual=None, ln=None, ln_eps=1e-5, gelu=False, gelu_in=False, out_f32=False,
               out=None, ln_stats=None):
    """act(LN?(a) @ weight^T + bias) + residual.   a: [..., K]; weight: [N, K] in a's dtype; bias / ln fp32.
    ln_stats = (mean, rstd) from layernorm_stats(a): the prologue then skips its own two passes over a."""
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
    mean = rstd = None
    if ln_stats is not None:
        mean, rstd = ln_stats
        assert ln is not None and mean.numel() == M and rstd.numel() == M
    # (half operands: the precise fused mode's unit of the same kernel, include/wmz.h)
    L.call('wmz_linear_fwd_stats' + ('_f16' if a.dtype == torch.float16 else ''), L.ptr(a), lda, L.ptr(weight), L.ptr(bias), L.ptr(residual), ldr, L.ptr(out), ldc,
           M, N, K, L.ptr(g), L.ptr(b), L.ptr(mean), L.ptr(rstd), float(ln_eps),
           (L.WMZ_LIN_GELU if gelu else 0) | (L.WMZ_LIN_GELU_IN if gelu_in else 0), 1 if out_f32 else 0, dt,
           L.stream())
    return out


KEEP_NORM_MIN_ROWS = 16384    # from here on the training forward's PreNorm GEMMs also write LN(a): the weight gradients then read
                              # a plain operand and take the 256-wide tiles (fewer rows: they sit on the graph's side branch anyway)


def linear_fwd_train(a, weight, bias, ln, ln_eps=1e-5, ln_stats=None, want_gelu=False, want_norm=False):
    """The training forward's PreNorm GEMM (wmz_linear_fwd_train): c = LN(a) @ weight^T + bias, and on request h = GELU(c) and
    an = LN(a) as the GEMM consumed it -> (c, h | None, an | None)."""
    K = a.shape[-1]
    N = weight.shape[0]
    dt = L.dtype_code(a.dtype)
    assert weight.dtype == a.dtype and weight.is_contiguous() and weight.shape[1] == K and ln is not None
    a, M, lda = _rows(a)
    lead = a.shape[:-1]
    c = torch.empty(lead + (N,), dtype=a.dtype, device=a.device)
    h = torch.empty(lead + (N,), dtype=a.dtype, device=a.device) if want_gelu else None
    an = torch.empty(lead + (K,), dtype=a.dtype, device=a.device) if want_norm else None
    g, b = ln
    mean = rstd = None
    if ln_stats is not None:
        mean, rstd = ln_stats
        assert mean.numel() == M and rstd.numel() == M
    assert bias is None or bias.dtype == torch.float32
    L.call('wmz_linear_fwd_train', L.ptr(a), lda, L.ptr(weight), L.ptr(bias), L.ptr(c), N, L.ptr(h), N, L.ptr(an), K, M, N, K,
           L.ptr(g), L.ptr(b), L.ptr(mean), L.ptr(rstd), float(ln_eps), dt, L.stream())
    return c, h, an


def linear_fwd_gelu_pair(a, weight, bias=None, ln=None, ln_eps=1e-5, ln_stats=None):
    """(z, h) = (LN?(a) @ weight^T + bias, GELU(z)), both in a's dtype, from one launch (wmz_linear_fwd_gelu_pair)."""
    K = a.shape[-1]
    N = weight.shape[0]
    dt = L.dtype_code(a.dtype)
    assert weight.dtype == a.dtype and weight.is_contiguous() and weight.shape[1] == K
    a, M, lda = _rows(a)
    lead = a.shape[:-1]
    z = torch.empty(lead + (N,), dtype=a.dtype, device=a.device)
    h = torch.empty(lead + (N,), dtype=a.dtype, device=a.device)
    g = b = mean = rstd = None
    if ln is not None:
        g, b = ln
        assert g.dtype == torch.float32 and b.dtype == torch.float32
    if ln_stats is not None:
        mean, rstd = ln_stats
        assert ln is not None and mean.numel() == M and rstd.numel() == M
    assert bias is None or bias.dtype == torch.float32
    L.call('wmz_linear_fwd_gelu_pair', L.ptr(a), lda, L.ptr(weight), L.ptr(bias), L.ptr(z), N, L.ptr(h), N, M, N, K,
           L.ptr(g), L.ptr(b), L.ptr(mean), L.ptr(rstd), float(ln_eps), dt, L.stream())
    return z, h


def linear_fwd_blocks(a, weight, bias=None, out_f32=False):
    """a: [Bk, R, K] with contiguous rows inside each block and an arbitrary block stride (x[:, -1] of a [B,S,H,W,D]
    stream, flattened to [B, H*W, D]) -> [Bk, R, N] = a @ weight^T + bias, the blocks read in place."""
    Bk, R, K = a.shape
    N = weight.shape[0]
    assert a.stride(2) == 1 and weight.dtype == a.dtype and weight.is_contiguous() and weight.shape[1] == K
    out = torch.empty((Bk, R, N), dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    L.call('wmz_linear_fwd_blocked' + ('_f16' if a.dtype == torch.float16 else ''), L.ptr(a), a.stride(1), R, a.stride(0), L.ptr(weight), L.ptr(bias), L.ptr(out), N,
           Bk * R, N, K, 1 if out_f32 else 0, L.dtype_code(a.dtype), L.stream())
    return out


def embed_pos3d_fwd(z, emb, pos_s, pos_h, pos_w, dtype):
    B, S, H, W = z.shape
    D = emb.shape[1]
    z = z.contiguous()
    x = torch.empty((B, S, H, W, D), dtype=dtype, device=z.device)
    L.call('wmz_embed_pos3d_fwd', L.ptr(z), L.ptr(emb), L.ptr(pos_s), L.ptr(pos_h), L.ptr(pos_w), L.ptr(x),
           B, S, H, W, D, emb.shape[0], L.dtype_code(dtype), L.stream())
    return x


_vq_screen_ws = {}      # device -> byte scratch of the screened nearest-code search (grown on demand)


def vq_argmin(x, codebook, need_dist=False, exact_scan=False):
    """x: [N,E] fp32, codebook: [C,E] fp32 -> int64 [N] (+ fp32 min distance), bit-identical to the reference's CPU result.
    embedding_dim 64 with a multiple of 64 codes: screening on the matrix cores + exact re-check (wmz_vq_argmin_screened);
    anything else, or exact_scan=True: every (row, code) distance in the pinned fp32 order (wmz_vq_argmin)."""
    assert x.dtype == torch.float32 and codebook.dtype == torch.float32
    x, N, ldx = _rows(x)
    codebook = codebook.contiguous()
    C, E = codebook.shape
    idx = torch.empty((N,), dtype=torch.int64, device=x.device)
    dmin = torch.empty((N,), dtype=torch.float32, device=x.device) if need_dist else None
    need = 0 if (exact_scan or N == 0 or ldx % 4 or x.data_ptr() % 16) else L.lib().wmz_vq_argmin_screened_workspace_bytes(N, C, E)
    if need > 0:
        ws = _vq_screen_ws.get(x.device)
        if ws is None or ws.numel() < need:
            ws = _vq_screen_ws[x.device] = torch.empty(need, dtype=torch.uint8, device=x.device)
        L.call('wmz_vq_argmin_screened', L.ptr(x), ldx, L.ptr(codebook), L.ptr(idx), L.ptr(dmin), N, C, E, L.ptr(ws), ws.numel(),
               L.stream())
    else:
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


_loss_partials = {}


def _partials(device):
    """fp32 workspace of the loss kernels' per-workgroup sums (one per device; the launches of a step run in stream order)."""
    w = _loss_partials.get(device)
    if w is None:
        w = _loss_partials[device] = torch.empty(int(L.lib().wmz_loss_partials_workspace_floats()), dtype=torch.float32, device=device)
    return w


class _VqTail(torch.autograd.Function):
    """vq.py:67-73 behind the nearest-code search: commitment loss, straight-through estimator, perplexity (wmz_vq_tail_fwd /
    _bwd).  inp: the quantiser's input as the caller holds it (any float dtype, [..., E]); flat / q: fp32 [N, E] (the input's fp32
    rows the search ran on, the gathered codebook rows); counts fp32 [C].  -> (straight-through tensor [..., Ep] in out_dtype,
    loss, perplexity)."""

    @staticmethod
    def forward(ctx, inp, flat, q, counts, out_dtype, Ep):
        N, E = flat.shape
        dev = flat.device
        st = torch.empty(inp.shape[:-1] + (Ep,), dtype=out_dtype, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ppl = torch.empty((), dtype=torch.float32, device=dev)
        L.call('wmz_vq_tail_fwd', L.ptr(flat), L.ptr(q), L.ptr(counts), L.ptr(st), L.ptr(_partials(dev)), L.ptr(loss), L.ptr(ppl),
               N, E, Ep, counts.numel(), L.dtype_code(out_dtype), L.stream())
        ctx.save_for_backward(flat, q)
        ctx.meta = (inp.shape, inp.dtype, Ep, out_dtype)
        ctx.mark_non_differentiable(ppl)
        return st, loss, ppl

    @staticmethod
    def backward(ctx, d_st, d_loss, _d_ppl):
        flat, q = ctx.saved_tensors
        shape, in_dtype, Ep, out_dtype = ctx.meta
        N, E = flat.shape
        if d_st is not None:
            d_st = d_st.contiguous()
            assert d_st.dtype == out_dtype
        if d_loss is not None:
            d_loss = d_loss.reshape(1).float()
        dx = torch.empty(shape, dtype=in_dtype, device=flat.device)
        L.call('wmz_vq_tail_bwd', L.ptr(d_st), L.ptr(flat), L.ptr(q), L.ptr(d_loss), L.ptr(dx), N, E, Ep, L.dtype_code(in_dtype),
               L.dtype_code(out_dtype), L.stream())
        return dx, None, None, None, None, None


def vq_tail(inp, flat, q, counts, out_dtype, Ep):
    return _VqTail.apply(inp, flat, q, counts, out_dtype, int(Ep))


RECON_LOSS_KINDS = {'SmoothL1': 0, 'MSE': 1, 'MAE': 2, 'L1': 2}


class _ReconLoss(torch.autograd.Function):
    """mean loss(y[..., :C] - target) with y the decoder's NHWC output [B, H, W, Cp] (padding channels ignored) and target the
    NCHW fp32 frames [B, C, H, W] (wmz_recon_loss_fwd / _bwd: train_vqae.py:139-150)."""

    @staticmethod
    def forward(ctx, y, target, kind):
        B, H, W, Cp = y.shape
        C = target.shape[1]
        assert y.is_contiguous() and target.is_contiguous() and target.dtype == torch.float32 and target.shape == (B, C, H, W)
        loss = torch.empty((), dtype=torch.float32, device=y.device)
        L.call('wmz_recon_loss_fwd', L.ptr(y), L.ptr(target), L.ptr(_partials(y.device)), L.ptr(loss), B, H * W, C, Cp, kind,
               L.dtype_code(y.dtype), L.stream())
        ctx.save_for_backward(y, target)
        ctx.kind = kind
        return loss

    @staticmethod
    def backward(ctx, g):
        y, target = ctx.saved_tensors
        B, H, W, Cp = y.shape
        dy = torch.empty_like(y)
        L.call('wmz_recon_loss_bwd', L.ptr(y), L.ptr(target), L.ptr(g.reshape(1).float()), L.ptr(dy), B, H * W, target.shape[1], Cp,
               ctx.kind, L.dtype_code(y.dtype), L.stream())
        return dy, None, None


def recon_loss(y_nhwc, target_nchw, kind):
    return _ReconLoss.apply(y_nhwc, target_nchw, RECON_LOSS_KINDS[kind])


def sample_tokens(logits, top_k, alphas, mask_token, last_frame, denoised, counter, seed, last_mask=None):
    """wmz_sample_tokens_dev: logits fp32 [R, C]; last_frame: the [B, H, W] int64 view batch_z[:, -1] the tokens are written
    into in place; denoised int64 [R]; alphas fp32 [n]; counter int64 [1] (device); last_mask uint8 [R] or None."""
    R, C = logits.shape
    B = last_frame.shape[0]
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and last_frame.dtype == torch.int64
    assert last_frame[0].is_contiguous() and R % B == 0 and denoised.numel() == R and counter.dtype == torch.int64
    L.call('wmz_sample_tokens_dev', L.ptr(logits), logits.stride(0), R, C, int(top_k), L.ptr(alphas), alphas.numel(), int(mask_token),
           L.ptr(last_frame), R // B, last_frame.stride(0), L.ptr(denoised), L.ptr(last_mask), int(seed) & 0xFFFFFFFFFFFFFFFF,
           L.ptr(counter), L.stream())


_vq_ws = {}


def vq_ema_stats(x, idx, codebook, counts=None, dw=None, sqerr=None):
    """counts[c] += #rows of code c, dw[c] += their sum, sqerr[c] += their squared distance to the code (vq.py:35-36, :43-46):
    rows counting-sorted by code, then gathered (no atomic per element; the scatter kernel stays for > 12 288 codes)."""
    x, N, ldx = _rows(x)
    C, E = codebook.shape
    idx = idx.reshape(-1).contiguous()
    if C <= 12288 and E <= 256:
        need = L.lib().wmz_vq_ema_stats_workspace_ints(N, C)
        ws = _vq_ws.get(x.device)
        if ws is None or ws.numel() < need:
            ws = _vq_ws[x.device] = torch.zeros(need, dtype=torch.int32, device=x.device)     # zeroed ONCE: calls re-zero the counters
        L.call('wmz_vq_ema_stats_sorted', L.ptr(x), ldx, L.ptr(idx), L.ptr(codebook), L.ptr(counts), L.ptr(dw), L.ptr(sqerr),
               N, C, E, L.ptr(ws), ws.numel(), L.stream())
        return
    L.call('wmz_vq_ema_stats', L.ptr(x), ldx, L.ptr(idx), L.ptr(codebook), L.ptr(counts), L.ptr(dw), L.ptr(sqerr), N, C, E,
           L.stream())


def vq_ema_update(embedding, cluster_size, activation_count, counts, dw, decay, eps):
    C, E = embedding.shape[-2:]
    L.call('wmz_vq_ema_update', L.ptr(embedding), L.ptr(cluster_size), L.ptr(activation_count), L.ptr(counts),
           L.ptr(dw), C, E, float(decay), float(eps), L.stream())


# ------------------------------------------------------------------------------------------------ backward

def local3d_attention_bwd(q, k, v, out, lse, dout, extents, heads, dqkv=None):
    """Returns (dq, dkv) with dkv = [..., 2I] holding dk | dv (the layout the fused k|v projection wants).
    dqkv (optional, [..., 3I]): write dq | dk | dv into its column thirds instead (fused to_qkv of config 5)."""
    B, S, H, W, I = q.shape
    dh = I // heads
    dt = L.dtype_code(q.dtype)
    q, _, ldq = _rows(q)
    k, _, ldk = _rows(k)
    v, _, ldv = _rows(v)
    out, _, ldo = _rows(out)
    dout, _, lddo = _rows(dout)
    delta = torch.empty((2, B * S * H * W, heads), dtype=torch.float32, device=q.device)   # delta | -lse / scale (wmz.h)
    if dqkv is None:
        dq = torch.empty((B, S, H, W, I), dtype=q.dtype, device=q.device)
        dkv = torch.empty((B, S, H, W, 2 * I), dtype=q.dtype, device=q.device)
        dk, dv = dkv[..., :I], dkv[..., I:]
        lddq, lddkv = I, 2 * I
    else:
        assert dqkv.is_contiguous() and dqkv.shape[-1] == 3 * I and dqkv.dtype == q.dtype
        dq, dk, dv = dqkv[..., :I], dqkv[..., I:2 * I], dqkv[..., 2 * I:]
        dkv = dqkv[..., I:]
        lddq = lddkv = 3 * I
    L.call('wmz_local3d_attn_bwd', L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(dout), L.ptr(dq),
           L.ptr(dk), L.ptr(dv), L.ptr(delta), B, S, H, W, heads, dh, int(extents[0]), int(extents[1]),
           int(extents[2]), ldq, ldk, ldv, ldo, lddo, lddq, lddkv, lddkv, dt, L.stream())
    return dq, dkv


def linear_dgrad(dc, weight_t, dgelu_z=None):
    """dA' = dC @ W (weight_t = W^T contiguous [K, N] in dC's dtype), optionally times gelu'(z)."""
    K = weight_t.shape[0]
    dt = L.dtype_code(dc.dtype)
    dc, M, ldc = _rows(dc)
    N = dc.shape[-1]
    out = torch.empty(dc.shape[:-1] + (K,), dtype=dc.dtype, device=dc.device)
    ldz = 0
    if dgelu_z is not None:
        dgelu_z, Mz, ldz = _rows(dgelu_z)
        assert Mz == M
    L.call('wmz_linear_fwd', L.ptr(dc), ldc, L.ptr(weight_t), None, L.ptr(dgelu_z), ldz, L.ptr(out), K, M, K, N,
           None, None, 0.0, L.WMZ_LIN_DGELU if dgelu_z is not None else 0, 0, dt, L.stream())
    return out


def workspace_snapshot(device):
    """The per-device library workspaces as they are now (strong references): a captured hipGraph bakes their addresses in,
    so its owner holds this tuple: when ops later replaces one by a larger allocation, the graph goes on using the one it holds."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    side = _wgrad_side.get(device)
    return (_wgrad_ws.get(device), _embed_ws.get(device), _vq_ws.get(device), _vq_screen_ws.get(device),
            None if side is None else side[2])


def workspace_same(a, b):
    return all(x is y for x, y in zip(a, b))


_wgrad_ws = {}          # device -> fp32 scratch for the two-stage weight-gradient reduction (grown on demand, never shrunk)
# at least 128 MB: the 256-wide tiles (wgrad3_kernel) write one 256 KB partial tile per workgroup, ~one workgroup per CU -- up to
# ~70 MB per launch whatever the problems are; the library falls back to the 128-wide tiles when the workspace is smaller
WGRAD_WS_MIN_FLOATS = 1 << 25


def _workspace(device, floats):
    w = _wgrad_ws.get(device)
    if w is None or w.numel() < floats:
        w = torch.empty(max(floats, WGRAD_WS_MIN_FLOATS), dtype=torch.float32, device=device)
        _wgrad_ws[device] = w
    return w


# ---- weight gradients on a side stream ------------------------------------------------------------------------------
# A weight gradient that goes straight into the flat gradient arena (backward._emit*) has ONE consumer, the optimizer: nothing
# of the backward's dependency chain (dgrad -> attention backward -> dgrad ...) waits for it.  With config.wgrad_stream on,
# those launches go to a per-device side stream that forks from the compute stream at the launch (its operands are complete
# there) and joins it again when the autograd pass ends (an engine callback; under hipGraph capture these are the fork / join
# edges of the graph).  At small token counts (config 5: 3 072 rows per GPU) every kernel of the backward is a partial wave of
# latency-bound workgroups, and the weight gradients -- a third of the step's kernel time -- run beside the chain instead of
# inside it.  The data-parallel reducer waits for this stream as well before it all-reduces a bucket (parallel._launch).
_wgrad_side = {}          # device -> [stream, launches pending a join, its own reduction workspace, operands held until the join]
_arena_depth = 0          # > 0 inside backward._emit*: the gradient being produced lands in the arena
_side_blocked = 0         # > 0: a block backward whose incoming gradient is handed on to autograd as it is (see side_blocked)


@contextlib.contextmanager
def arena_fill():
    global _arena_depth
    _arena_depth += 1
    try:
        yield
    finally:
        _arena_depth -= 1


@contextlib.contextmanager
def side_blocked(block=True):
    """A block backward that returns its incoming gradient tensor unchanged (an un-folded residual path) keeps its weight
    gradients on the compute stream: autograd may accumulate INTO that tensor in place right after the function returns,
    while a side-stream weight gradient would still be reading it."""
    global _side_blocked
    _side_blocked += 1 if block else 0
    try:
        yield
    finally:
        _side_blocked -= 1 if block else 0


# A side-stream launch is ISSUED only after the compute stream's next library launch: under hipGraph capture the order in
# which a node's successors are recorded decides which of them the runtime keeps on the node's own queue -- recorded first,
# the weight gradient stayed there and the backward's dependency chain hopped to another queue at every fork (a ~10 us
# cross-queue hand-over each, and the chain's next node queued behind the weight gradient: measured on config 5).
_deferred = []            # launches to issue behind the compute stream's next library call
_pending = []             # plain / LayerNorm-prologue weight gradients waiting to leave as ONE batched launch pair
_pending_conv = []        # the same for the conv layers' weight gradients (wmz_conv2d_nhwc_wgrad_batch)
_flushing = False
WGRAD_BATCH = 6           # problems per wmz_linear_wgrad_batch_ln call (the library's limit)


def linear_wgrad_batch_ln(problems, side=None):
    """Up to 6 weight gradients, each with an optional LayerNorm prologue, by one launch pair (wmz_linear_wgrad_batch_ln).
    problems: dicts with dc, a (2-D views, row strides ldc / lda), dw, dbias | None, M, N, K, g, b, mean, rstd (None = plain),
    overwrite.  side: the side-stream entry whose workspace to use."""
    import ctypes
    n = len(problems)
    vp, ci, cl = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_long * n
    pc, pa, pw, pb, pg, pbe, pm, pr = vp(), vp(), vp(), vp(), vp(), vp(), vp(), vp()
    lc, la, Ms, Ns, Ks, ov = cl(), cl(), ci(), ci(), ci(), ci()
    need = 0
    dt = L.dtype_code(problems[0]['dc'].dtype)
    for i, q in enumerate(problems):
        assert L.dtype_code(q['dc'].dtype) == dt and q['a'].dtype == q['dc'].dtype and q['dw'].dtype == torch.float32
        pc[i], pa[i], pw[i], pb[i] = L.ptr(q['dc']), L.ptr(q['a']), L.ptr(q['dw']), L.ptr(q['dbias'])
        pg[i], pbe[i], pm[i], pr[i] = L.ptr(q['g']), L.ptr(q['b']), L.ptr(q['mean']), L.ptr(q['rstd'])
        lc[i], la[i], Ms[i], Ns[i], Ks[i], ov[i] = q['ldc'], q['lda'], q['M'], q['N'], q['K'], 1 if q['overwrite'] else 0
        need += L.lib().wmz_linear_wgrad_workspace_floats(q['M'], q['N'], q['K'], dt)
    dev = problems[0]['dc'].device
    ws = _workspace(dev, need) if side is None else _side_workspace(side, dev, need)
    L.call('wmz_linear_wgrad_batch_ln', n, pc, lc, pa, la, pw, pb, Ms, Ns, Ks, ov, pg, pbe, pm, pr, L.ptr(ws), ws.numel(), dt,
           L.stream())


def _issue_pending_conv():
    """The queued conv weight gradients as one launch pair on the side stream, behind the LAST one's fork point."""
    global _pending_conv
    if not _pending_conv:
        return
    import ctypes
    batch, _pending_conv = _pending_conv[:WGRAD_BATCH], _pending_conv[WGRAD_BATCH:]
    n = len(batch)
    vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
    px, pdy, pw, pb = vp(), vp(), vp(), vp()
    cols = [ci() for _ in range(12)]
    need = 0
    for i, q in enumerate(batch):
        px[i], pdy[i], pw[i], pb[i] = L.ptr(q['x']), L.ptr(q['dy']), L.ptr(q['gw']), L.ptr(q['gb'])
        for c, v in zip(cols, q['geom']):
            c[i] = v
        need += q['need']
    ent = batch[0]['ent']
    ent[0].wait_event(batch[-1]['fork'])
    with torch.cuda.stream(ent[0]):
        ws = _side_workspace(ent, batch[0]['x'].device, need)
        L.call('wmz_conv2d_nhwc_wgrad_batch', n, px, pdy, pw, pb, *cols, L.ptr(ws), ws.numel(), batch[0]['dt'], L.stream())
    if _pending_conv:
        _issue_pending_conv()


def _issue_pending():
    """The queued weight gradients as one launch pair on the side stream (behind the LAST one's fork point: the compute stream
    is in order, so the earlier operands are complete there too).  At few tokens the captured step is a chain of launches each
    of which the host has to enqueue (~9 us a graph node on this stack) and the device to start: config 5's 32 weight gradients
    x 2 launches become 6 launches."""
    global _pending
    if not _pending:
        return
    batch, _pending = _pending, []
    ent = batch[0]['ent']
    ent[0].wait_event(batch[-1]['fork'])
    with torch.cuda.stream(ent[0]):
        linear_wgrad_batch_ln(batch, side=ent)
        for q in batch:
            if q['then'] is not None:
                q['then']()


def _flush_deferred(force=True):
    global _flushing
    if _flushing or not (_deferred or _pending or _pending_conv):
        return
    _flushing = True
    try:
        while _deferred:
            _deferred.pop(0)()
        if force or len(_pending) >= WGRAD_BATCH:
            _issue_pending()
        if force or len(_pending_conv) >= WGRAD_BATCH:
            _issue_pending_conv()
    finally:
        _flushing = False
        if not (_deferred or _pending or _pending_conv):
            L.after_call = None


def _after_call():
    _flush_deferred(force=False)


DEFER_SIDE = int(__import__('os').environ.get('WMZ_WGRAD_DEFER', '1'))      # development: 0 = record side launches at their fork


def _defer(launch):
    if _flushing or not DEFER_SIDE:    # (a deferred launch's own L.call)
        launch()
        return
    _deferred.append(launch)
    L.after_call = _after_call


def wgrad_join():
    """The current stream waits for every weight gradient launched on the side stream (no-op when there is none)."""
    _flush_deferred()
    for dev, ent in _wgrad_side.items():
        if ent[1]:
            torch.cuda.current_stream(dev).wait_stream(ent[0])
            ent[1] = False
        # only now may the operands go: until the join the side stream may still be READING them, and while this list holds a
        # reference autograd cannot take a gradient buffer for an in-place accumulation (its storage is shared)
        del ent[3][:]


def wgrad_join_at_end():
    """Called inside an autograd pass: the pass ends with wgrad_join() (for side-stream launches forked BEFORE the pass began)."""
    torch.autograd.Variable._execution_engine.queue_callback(wgrad_join)


def wgrad_reset():
    """Forget every queued / deferred side-stream launch and the bookkeeping around them: called before a step is captured and
    after a capture or a backward pass failed, so that nothing stale is flushed into the next step."""
    global _pending, _pending_conv, _arena_depth, _side_blocked, _flushing
    _pending = []
    _pending_conv = []
    del _deferred[:]
    _arena_depth = _side_blocked = 0
    _flushing = False
    L.after_call = None
    for ent in _wgrad_side.values():
        ent[1] = False
        del ent[3][:]


def wgrad_side_wait(stream):
    """`stream` (the reducer's) waits for the side-stream weight gradients launched so far; they stay pending for the join."""
    _flush_deferred()
    for ent in _wgrad_side.values():
        if ent[1]:
            stream.wait_stream(ent[0])


def _wgrad_side_enter(device, tensors, join_later=False):
    """None (launch on the current stream), or the side-stream entry: forked from the current stream, `tensors` (allocated on
    the compute stream) marked as in use there.  join_later: the caller is NOT inside an autograd pass (a gradient produced by a
    forward: train._LinearCrossEntropy) and promises that one follows, whose first node calls wgrad_join_at_end()."""
    from . import config
    # (hipGraph capture only: in eager launches the step is host-bound and the extra stream / event calls cost more than the
    #  overlap returns -- config 5: 4.8 -> 5.9 ms per step eager, 4.06 -> 3.38 as a graph)
    if (_arena_depth == 0 or _side_blocked > 0 or not config.get_wgrad_stream()
            or not torch.cuda.is_current_stream_capturing()):
        return None
    if not join_later:
        try:
            # (one callback per launch, each a no-op once the first has joined: no state that a failed pass could leave behind)
            torch.autograd.Variable._execution_engine.queue_callback(wgrad_join)
        except RuntimeError:            # not inside an autograd pass (a block backward called by hand): stay on the stream
            return None
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    ent = _wgrad_side.get(device)
    if ent is None:
        ent = _wgrad_side[device] = [torch.cuda.Stream(device=device), False, None, []]
    for t in tensors:
        if t is not None:
            t.record_stream(ent[0])
            ent[3].append(t)                # alive (and not uniquely owned) until wgrad_join()
    ent[1] = True
    fork = torch.cuda.Event()
    fork.record(torch.cuda.current_stream(device))      # the operands are complete here
    return ent, fork


def _side_workspace(ent, device, floats):
    w = ent[2]
    if w is None or w.numel() < floats:
        w = ent[2] = torch.empty(max(floats, WGRAD_WS_MIN_FLOATS), dtype=torch.float32, device=device)
    return w


def linear_wgrad(dc, a, dw, dbias=None, ln=None, ln_stats=None, gelu_in=False, overwrite=False, then=None, join_later=False):
    """dw[N,K] += dc^T @ a' ; dbias[N] += colsum(dc) (overwrite: = instead of +=).  dw / dbias fp32 (two-stage reduction
    through a per-device workspace: deterministic, no float atomics).  then(): follow-up work on the result, issued behind the
    launch on whichever stream it went to."""
    dt = L.dtype_code(dc.dtype)
    dc, M, ldc = _rows(dc)
    a, Ma, lda = _rows(a)
    assert Ma == M and a.dtype == dc.dtype and dw.dtype == torch.float32 and dw.is_contiguous()
    N, K = dw.shape
    g = b = mean = rstd = None
    if ln is not None:
        g, b = ln
        mean, rstd = ln_stats
    need = L.lib().wmz_linear_wgrad_workspace_floats(M, N, K, dt)
    side = _wgrad_side_enter(dc.device, (dc, a, g, b, mean, rstd, dbias), join_later)
    if side is None:
        ws = _workspace(dc.device, need)
        L.call('wmz_linear_wgrad_ws', L.ptr(dc), ldc, L.ptr(a), lda, L.ptr(dw), L.ptr(dbias), M, N, K, L.ptr(g), L.ptr(b),
               L.ptr(mean), L.ptr(rstd), 1 if gelu_in else 0, 1 if overwrite else 0, L.ptr(ws), ws.numel(), dt, L.stream())
        if then is not None:
            then()
        return
    ent, fork = side
    if not gelu_in:
        # batchable: queued (the queue keeps the operands alive), issued when WGRAD_BATCH problems wait or at the join
        if _pending and (_pending[0]['dt'] != dt or _pending[0]['ent'] is not ent or len(_pending) >= WGRAD_BATCH):
            _flush_deferred()
        _pending.append(dict(dc=dc, ldc=ldc, a=a, lda=lda, dw=dw, dbias=dbias, M=M, N=N, K=K, g=g, b=b, mean=mean, rstd=rstd,
                             overwrite=overwrite, need=need, dt=dt, ent=ent, fork=fork, then=then))
        L.after_call = _after_call
        return

    def launch():                       # (the closure keeps the operands alive until the launch has been issued)
        ent[0].wait_event(fork)
        with torch.cuda.stream(ent[0]):
            ws = _side_workspace(ent, dc.device, need)
            L.call('wmz_linear_wgrad_ws', L.ptr(dc), ldc, L.ptr(a), lda, L.ptr(dw), L.ptr(dbias), M, N, K, L.ptr(g), L.ptr(b),
                   L.ptr(mean), L.ptr(rstd), 1 if gelu_in else 0, 1 if overwrite else 0, L.ptr(ws), ws.numel(), dt, L.stream())
            if then is not None:
                then()
    _defer(launch)


def side_branch(device, tensors, work):
    """Run work() -- library launches whose results only the optimizer (and the data-parallel reducer) consume: a layer's batched
    weight gradients -- on the weight-gradient side stream when a step is being captured (forked where `tensors`, its operands,
    are complete; issued behind the compute stream's next launch; joined when the autograd pass ends), else right here.
    work(side): side = the side-stream entry (its workspace) or None."""
    with arena_fill():
        side = _wgrad_side_enter(device, tensors)
    if side is None:
        work(None)
        return
    ent, fork = side

    def launch():
        ent[0].wait_event(fork)
        with torch.cuda.stream(ent[0]):
            work(ent)
    _defer(launch)


def linear_wgrad_batch(problems, side=None):
    """Up to 6 plain weight gradients by one launch pair (wmz_linear_wgrad_batch).  problems: (dc, a, dw, dbias | None,
    overwrite[, a_tiled]) tuples with the meaning of linear_wgrad's arguments; a_tiled: `a` is the fused path's tiled stream
    ([M, 256] bf16 in 32-row tiles) instead of row-major.  (Stays on the compute stream: at config 4 these launches fill the
    chip, a side branch measured 1.91 vs 1.90 ms per step.)"""
    import ctypes
    n = len(problems)
    dt = L.dtype_code(problems[0][0].dtype)
    vp, ci, cl = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_long * n
    pc, pa, pw, pb, lc, la, Ms, Ns, Ks, ov, tl = vp(), vp(), vp(), vp(), cl(), cl(), ci(), ci(), ci(), ci(), ci()
    keep, need = [], 0
    for i, prob in enumerate(problems):
        dc, a, dw, dbias, overwrite = prob[:5]
        tiled = len(prob) > 5 and bool(prob[5])
        dc, M, ldc = _rows(dc)
        if tiled:
            assert a.is_contiguous() and a.numel() == M * 256 and a.dtype == torch.bfloat16
            Ma, lda = M, 256
        else:
            a, Ma, lda = _rows(a)
        assert Ma == M and a.dtype == dc.dtype and L.dtype_code(dc.dtype) == dt
        assert dw.dtype == torch.float32 and dw.is_contiguous()
        N, K = dw.shape
        keep.append((dc, a))
        pc[i], pa[i], pw[i], pb[i] = L.ptr(dc), L.ptr(a), L.ptr(dw), L.ptr(dbias)
        lc[i], la[i], Ms[i], Ns[i], Ks[i], ov[i], tl[i] = ldc, lda, M, N, K, 1 if overwrite else 0, 1 if tiled else 0
        need += L.lib().wmz_linear_wgrad_workspace_floats(M, N, K, dt)
    dev = problems[0][0].device
    ws = _workspace(dev, need) if side is None else _side_workspace(side, dev, need)
    L.call('wmz_linear_wgrad_batch', n, pc, lc, pa, la, pw, pb, Ms, Ns, Ks, ov, tl, L.ptr(ws), ws.numel(), dt, L.stream())


def layernorm_stats(x, eps=1e-5):
    x, M, ldx = _rows(x)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    L.call('wmz_layernorm_stats', L.ptr(x), ldx, L.ptr(mean), L.ptr(rstd), M, x.shape[-1], float(eps),
           L.dtype_code(x.dtype), L.stream())
    return mean, rstd


def layernorm_bwd(x, dyhat, gamma, dgamma, dbeta, skip=None, eps=1e-5, skip2=None):
    x, M, ldx = _rows(x)
    dyhat, _, lddy = _rows(dyhat)
    lds = lds2 = 0
    if skip is None and skip2 is not None:
        skip, skip2 = skip2, None
    if skip is not None:
        skip, _, lds = _rows(skip)
    if skip2 is not None:
        skip2, _, lds2 = _rows(skip2)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    K = x.shape[-1]
    L.call('wmz_layernorm_bwd', L.ptr(x), ldx, L.ptr(dyhat), lddy, L.ptr(skip), lds, L.ptr(skip2), lds2, L.ptr(gamma), L.ptr(dx), K,
           L.ptr(dgamma), L.ptr(dbeta), M, K, float(eps), L.dtype_code(x.dtype), L.stream())
    return dx


_embed_ws = {}


def embed_pos3d_bwd(z, dx, tabs):
    """Accumulates into tabs = [demb, dpos_s, dpos_h, dpos_w] (fp32).  The denoiser's shapes (W = 16, D = 256, bf16) take the
    counting-sort + gather path (csrc/embed_bwd.hip: no float atomic per element), everything else the scatter kernels."""
    B, S, H, W = z.shape
    D = dx.shape[-1]
    dx = dx.contiguous()
    C = tabs[0].shape[0]
    if W == 16 and D == 256 and dx.dtype == torch.bfloat16 and C <= 12288:
        need = L.lib().wmz_embed_pos3d_bwd_workspace_ints(B, S, H, W, C)
        ws = _embed_ws.get(dx.device)
        if ws is None or ws.numel() < need:
            ws = _embed_ws[dx.device] = torch.zeros(need, dtype=torch.int32, device=dx.device)    # zeroed ONCE: calls re-zero the counters
        L.call('wmz_embed_pos3d_bwd_sorted', L.ptr(z.contiguous()), L.ptr(dx), L.ptr(tabs[0]), L.ptr(tabs[1]), L.ptr(tabs[2]),
               L.ptr(tabs[3]), B, S, H, W, D, C, L.ptr(ws), ws.numel(), L.dtype_code(dx.dtype), L.stream())
        return tabs
    L.call('wmz_embed_pos3d_bwd', L.ptr(z.contiguous()), L.ptr(dx), L.ptr(tabs[0]), L.ptr(tabs[1]), L.ptr(tabs[2]),
           L.ptr(tabs[3]), B, S, H, W, D, C, L.dtype_code(dx.dtype), L.stream())
    return tabs


# ------------------------------------------------------------------------------------------------ conv AE (NHWC)

STAT_REPLICAS = 8          # include/wmz.h: WMZ_STAT_REPLICAS


DIRECT_CONV = True         # tools / tests: False keeps every convolution on the implicit-GEMM kernel (A/B comparisons)


def _direct_pack(w_op, Cin, Cout):
    """The fragment-order weight stream of csrc/conv_direct.hip for a GEMM operand [Cout, 9 * Cin], kept while the operand lives
    and the parameters have not been rewritten (the operand copies themselves are cached per parameter version: _cast)."""
    from . import _cast

    def build(w):
        n = L.lib().wmz_conv3x3_direct_pack_elems(Cin, Cout)
        dst = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        L.call('wmz_conv3x3_direct_pack', L.ptr(w), L.ptr(dst), Cin, Cout, L.stream())
        return dst
    return _cast.cached((w_op,), 'convq', build)


def _point_pack(w_op, K, Cout):
    """The fragment-order weights of csrc/conv_point.hip for a GEMM operand [Cout, K] (kept like _direct_pack's)."""
    from . import _cast

    def build(w):
        n = L.lib().wmz_conv_point_pack_elems(K, Cout)
        dst = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        L.call('wmz_conv_point_pack', L.ptr(w), L.ptr(dst), K, Cout, L.stream())
        return dst
    return _cast.cached((w_op,), 'convp', build)


def conv2d_nhwc(x, w_op, KH, KW, stride, pad, bias=None, scale=None, shift=None, residual=None, leaky=False,
                slope=0.01, stats=False, pre=None):
    """x: [B,H,W,Cin] contiguous (Cin % 8 == 0), w_op: [Cout, KH*KW*Cin] in x's dtype -> [B,Ho,Wo,Cout] (+ sum, sq: fp32
    [STAT_REPLICAS, Cout] partial sums, summed by bn_finalize)."""
    B, Hi, Wi, Cin = x.shape
    Cout = w_op.shape[0]
    assert x.is_contiguous() and w_op.is_contiguous() and w_op.shape[1] == KH * KW * Cin and w_op.dtype == x.dtype
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    s = q = None
    if stats:
        s, q = _stat_pair(Cout, x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == x.dtype
    if (DIRECT_CONV and x.dtype == torch.bfloat16 and KH == 3 and KW == 3 and pad == 1 and pre is None and 0.0 <= slope <= 1.0
            and L.lib().wmz_conv3x3_direct_supported_strided(Hi, Wi, Cin, Cout, stride)):
        # csrc/conv_direct.hip: the haloed patch by LDS-DMA, decoupled waves (same arithmetic as the implicit-GEMM kernel)
        L.call('wmz_conv3x3_direct_fwd_strided', L.ptr(x), L.ptr(_direct_pack(w_op, Cin, Cout)), L.ptr(out), L.ptr(bias), L.ptr(scale),
               L.ptr(shift), L.ptr(residual), L.ptr(s), L.ptr(q), B, Hi, Wi, Cin, Cout, stride, 1 if leaky else 0, float(slope),
               L.stream())
        return (out, s, q) if stats else out
    psc, psh, psl = pre if pre is not None else (None, None, 0.0)     # 1x1 only: LeakyReLU(x * psc + psh) on load
    point = (DIRECT_CONV and x.dtype == torch.bfloat16 and residual is None and 0.0 <= slope <= 1.0
             and L.lib().wmz_conv_point_supported(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad)
             # (the streaming kernel's input prologue is built for 1x1 layers of <= 128 channels: wider hidden planes keep
             #  the implicit-GEMM kernel's prologue)
             and (pre is None or (KH == 1 and KW == 1 and pad == 0 and Cin <= 128)))
    pst = None
    if isinstance(psc, BnLazy):                   # the prologue's BatchNorm as raw statistics: the streaming kernel finalises it
        if point and not psc.done:
            pst, psc, psh = psc.struct(), None, None
        else:
            psc, psh = psc.materialize()
    if point:
        # csrc/conv_point.hip: small-K layers (1x1, 2x2 / stride 2, the 3-channel conv_1) as a persistent streaming kernel
        L.call('wmz_conv_point_fwd_bn', L.ptr(x), L.ptr(_point_pack(w_op, KH * KW * Cin, Cout)), L.ptr(out), L.ptr(bias), L.ptr(scale),
               L.ptr(shift), L.ptr(s), L.ptr(q), L.ptr(psc), L.ptr(psh), ctypes.addressof(pst) if pst is not None else None, float(psl),
               B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, 1 if leaky else 0, float(slope), L.stream())
        return (out, s, q) if stats else out
    L.call('wmz_conv2d_nhwc_fwd_pre', L.ptr(x), L.ptr(w_op), L.ptr(out), L.ptr(bias), L.ptr(scale), L.ptr(shift),
           L.ptr(residual), L.ptr(s), L.ptr(q), L.ptr(psc), L.ptr(psh), float(psl), B, Hi, Wi, Cin, Cout, KH, KW, stride,
           pad, 1 if leaky else 0, float(slope), L.dtype_code(x.dtype), L.stream())
    return (out, s, q) if stats else out


_stat_arena = None         # [zero tensor [slots, 2 * STAT_REPLICAS * 128], next slot] while a stat_arena() is open


@contextlib.contextmanager
def stat_arena(slots=32):
    """One zero fill for the BatchNorm statistics of a whole encoder / decoder pass: inside the context the (sum, sumsq) pairs
    of conv2d_nhwc(stats=True) / channel_stats_nhwc are slices of ONE zeroed tensor (allocated at the first request) instead of a
    torch.zeros each -- ten ~5 us fills per frame-encoder call.  Re-entrant: an inner context shares the outer arena."""
    global _stat_arena
    if _stat_arena is not None:
        yield
        return
    _stat_arena = [None, 0, slots]
    try:
        yield
    finally:
        _stat_arena = None


def _stat_pair(C, device):
    a = _stat_arena
    if a is not None and C <= 128:
        if a[0] is None:
            a[0] = torch.zeros((a[2], 2 * STAT_REPLICAS * 128), dtype=torch.float32, device=device)
        if a[1] < a[2] and a[0].device == device:
            row = a[0][a[1]]
            a[1] += 1
            return row[:2 * STAT_REPLICAS * C].view(2, STAT_REPLICAS, C).unbind(0)
    return torch.zeros((2, STAT_REPLICAS, C), dtype=torch.float32, device=device).unbind(0)      # one fill for both


def nchw_to_nhwc8(x, dtype):
    """logical NCHW (contiguous) -> [B, H, W, C8] in `dtype`, channels zero-padded to a multiple of 8: one launch."""
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, (C + 7) // 8 * 8), dtype=dtype, device=x.device)
    L.call('wmz_nchw_to_nhwc8', L.ptr(x), L.ptr(y), B, C, H, W, L.dtype_code(x.dtype), L.dtype_code(dtype), L.stream())
    return y


def channel_stats_nhwc(x):
    C = x.shape[-1]
    M = x.numel() // C
    s, q = _stat_pair(C, x.device)
    L.call('wmz_channel_stats_nhwc', L.ptr(x), M, C, L.ptr(s), L.ptr(q), L.dtype_code(x.dtype), L.stream())
    return s, q


BN_LAZY = os.environ.get('WMZ_BN_LAZY', '1') != '0'      # A/B: 0 = a wmz_bn_finalize launch between every convolution and its consumer


class BnLazy:
    """A training-mode nn.BatchNorm2d as the raw batch statistics its producer left (include/wmz.h wmz_bn_stats): the kernel that
    applies the normalisation finalises it (scale / shift from the sums, running statistics moved by one of its workgroups).
    .scale / .shift (/ .mean / .rstd with want_stats) are filled by that launch; materialize() runs wmz_bn_finalize instead (for a
    consumer that takes plain per-channel arrays)."""

    def __init__(self, bn, s, q, count, want_stats=False):
        C = bn.num_features
        dev = s.device
        self.bn, self.s, self.q, self.count, self.C = bn, s, q, float(count), C
        buf = torch.empty((4 if want_stats else 2, C), dtype=torch.float32, device=dev)
        self.scale, self.shift = buf[0], buf[1]
        self.mean, self.rstd = (buf[2], buf[3]) if want_stats else (None, None)
        self.done = False

    def struct(self):
        """The ctypes struct for ONE consuming launch (keep it alive until the call returns)."""
        assert not self.done, 'a BnLazy is consumed once (the running statistics move once)'
        self.done = True
        bn = self.bn
        mom = bn.momentum if bn.momentum is not None else 0.1
        track = bn.running_mean is not None
        return L.BnStats(L.ptr(self.s), L.ptr(self.q), L.ptr(bn.weight.detach()) if bn.weight is not None else None,
                         L.ptr(bn.bias.detach()) if bn.bias is not None else None, L.ptr(bn.running_mean) if track else None,
                         L.ptr(bn.running_var) if track else None,
                         L.ptr(bn.num_batches_tracked) if (track and bn.num_batches_tracked is not None) else None,
                         L.ptr(self.scale), L.ptr(self.shift), L.ptr(self.mean), L.ptr(self.rstd), self.count, float(mom), float(bn.eps))

    def materialize(self):
        if not self.done:
            self.done = True
            bn = self.bn
            mom = bn.momentum if bn.momentum is not None else 0.1
            nbt = bn.num_batches_tracked if bn.num_batches_tracked is not None else None
            L.call('wmz_bn_finalize', L.ptr(self.s), L.ptr(self.q), self.count, L.ptr(bn.weight.detach()), L.ptr(bn.bias.detach()),
                   L.ptr(bn.running_mean), L.ptr(bn.running_var), float(mom), float(bn.eps), 1, L.ptr(self.scale), L.ptr(self.shift),
                   L.ptr(self.mean), L.ptr(self.rstd), self.C, L.ptr(nbt), L.stream())
        return self.scale, self.shift


def bn_lazy(bn, s, q, count, want_stats=False):
    """bn_finalize deferred into the consumer (BnLazy) when the module is in training mode with tracked statistics; otherwise -- or
    with WMZ_BN_LAZY=0 -- finalised now: the returned object has .scale / .shift (/ .mean / .rstd) either way."""
    lz = BnLazy(bn, s, q, count, want_stats)
    if not (BN_LAZY and bn.training and bn.running_mean is not None and bn.weight is not None):
        lz.materialize()
    return lz


def bn_finalize(bn, s, q, count, want_stats=False):
    """(scale, shift[, mean, rstd]) of an nn.BatchNorm2d; in training mode also updates its running statistics."""
    C = bn.num_features
    dev = bn.running_mean.device
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    shift = torch.empty(C, dtype=torch.float32, device=dev)
    training = bn.training or bn.running_mean is None
    mom = bn.momentum if bn.momentum is not None else 0.1
    mean = torch.empty(C, dtype=torch.float32, device=dev) if want_stats else None
    rstd = torch.empty(C, dtype=torch.float32, device=dev) if want_stats else None
    nbt = bn.num_batches_tracked if (training and bn.num_batches_tracked is not None) else None    # counted by the kernel
    L.call('wmz_bn_finalize', L.ptr(s), L.ptr(q), float(count), L.ptr(bn.weight.detach()), L.ptr(bn.bias.detach()),
           L.ptr(bn.running_mean), L.ptr(bn.running_var), float(mom), float(bn.eps), 1 if training else 0,
           L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(rstd), C, L.ptr(nbt), L.stream())
    return (scale, shift, mean, rstd) if want_stats else (scale, shift)


def affine_act_nhwc(a, sa=None, ta=None, b=None, sb=None, tb=None, leaky=False, slope=0.01):
    """y = act(a * sa + ta (+ b * sb + tb)).  sa / sb may be a BnLazy (ta / tb then None): its BatchNorm is finalised by this launch."""
    C = a.shape[-1]
    M = a.numel() // C
    y = torch.empty_like(a)
    dt = L.dtype_code(a.dtype)
    lazy = [t for t in (sa, sb) if isinstance(t, BnLazy) and not t.done]
    if lazy and not (L.lib().wmz_affine_act_bn_supported(C, dt) and (a.data_ptr() | (b.data_ptr() if b is not None else 0)) % 16 == 0):
        for t in lazy:
            t.materialize()
    sta = stb = None
    if isinstance(sa, BnLazy):
        if sa.done:
            sa, ta = sa.scale, sa.shift
        else:
            sta, sa, ta = sa.struct(), None, None
    if isinstance(sb, BnLazy):
        if sb.done:
            sb, tb = sb.scale, sb.shift
        else:
            stb, sb, tb = sb.struct(), None, None
    if sta is None and stb is None:
        L.call('wmz_affine_act_nhwc', L.ptr(a), L.ptr(sa), L.ptr(ta), L.ptr(b), L.ptr(sb), L.ptr(tb), L.ptr(y), M, C,
               1 if leaky else 0, float(slope), dt, L.stream())
    else:
        L.call('wmz_affine_act_nhwc_bn', L.ptr(a), L.ptr(sa), L.ptr(ta), ctypes.addressof(sta) if sta is not None else None, L.ptr(b),
               L.ptr(sb), L.ptr(tb), ctypes.addressof(stb) if stb is not None else None, L.ptr(y), M, C, 1 if leaky else 0, float(slope),
               dt, L.stream())
    return y


DILATE_KERNEL = True


def dilate_nhwc(dy, Hz, Wz, stride):
    """dz [B, Hz, Wz, C] with dz[:, ::stride, ::stride] = dy and zeros elsewhere (one pass: wmz_dilate_nhwc)."""
    B, Ho, Wo, C = dy.shape
    dy = dy.contiguous()
    if not DILATE_KERNEL or B > 65535 or Hz > 65535:     # (the kernel's grid.y / grid.z limits; A/B: tools/time_vqae_modes.py) the fill + strided copy of torch
        dz = torch.zeros((B, Hz, Wz, C), dtype=dy.dtype, device=dy.device)
        dz[:, 0:(Ho - 1) * stride + 1:stride, 0:(Wo - 1) * stride + 1:stride] = dy
        return dz
    dz = torch.empty((B, Hz, Wz, C), dtype=dy.dtype, device=dy.device)
    L.call('wmz_dilate_nhwc', L.ptr(dy), L.ptr(dz), B, Ho, Wo, C, Hz, Wz, stride, L.dtype_code(dy.dtype), L.stream())
    return dz


def bilinear2x_nhwc(x):
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
    L.call('wmz_bilinear2x_nhwc', L.ptr(x), L.ptr(y), B, H, W, C, L.dtype_code(x.dtype), L.stream())
    return y


def embed_indexed_fwd(tok, pos, emb, pos_s, pos_h, pos_w, shape, dtype):
    """tok, pos: int64 [..]; tables fp32 -> x [.., D]."""
    S, H, W = shape
    D = emb.shape[1]
    tok, pos = tok.contiguous(), pos.contiguous()
    x = torch.empty(tok.shape + (D,), dtype=dtype, device=tok.device)
    L.call('wmz_embed_indexed_fwd', L.ptr(tok), L.ptr(pos), L.ptr(emb), L.ptr(pos_s), L.ptr(pos_h), L.ptr(pos_w),
           L.ptr(x), tok.numel(), S, H, W, D, emb.shape[0], L.dtype_code(dtype), L.stream())
    return x


def embed_indexed_bwd(tok, pos, dx, shape, table_shapes, into=None):
    """-> the four table gradients (fp32).  into: four fp32 tensors of those shapes the kernel ACCUMULATES into instead (the
    parameters' slices of the flat gradient arena: no zero fills, no `grad += g` passes)."""
    S, H, W = shape
    dx = dx.contiguous()
    D = dx.shape[-1]
    if into is not None:
        tabs = list(into)
        assert all(t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(s) for t, s in zip(tabs, table_shapes))
    else:
        tabs = [torch.zeros(s, dtype=torch.float32, device=dx.device) for s in table_shapes]
    L.call('wmz_embed_indexed_bwd', L.ptr(tok.contiguous()), L.ptr(pos.contiguous()), L.ptr(dx), L.ptr(tabs[0]),
           L.ptr(tabs[1]), L.ptr(tabs[2]), L.ptr(tabs[3]), tok.numel(), S, H, W, D, table_shapes[0][0],
           L.dtype_code(dx.dtype), L.stream())
    return tabs


def conv2d_nhwc_wgrad(x, dy, KH, KW, stride, pad, want_bias, into=None):
    """x [B,Hi,Wi,Cin8], dy [B,Ho,Wo,Cout8] -> dW fp32 [Cout8, KH*KW*Cin8] (+ dbias fp32 [Cout8]).
    into = (weight_grad [Co, Ci, KH, KW], bias_grad [Co] | None): ACCUMULATE into nn.Conv2d's own gradient tensors instead
    (channel padding cropped, taps transposed by the reduction kernel); returns (None, None)."""
    B, Hi, Wi, Cin = x.shape
    Cout = dy.shape[-1]
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype
    dt = L.dtype_code(x.dtype)
    need = L.lib().wmz_conv2d_nhwc_wgrad_workspace_floats(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, dt)
    if into is not None:
        gw, gb = into
        co, ci = gw.shape[:2]
        assert gw.is_contiguous() and gw.dtype == torch.float32 and gw.shape[2:] == (KH, KW) and co <= Cout and ci <= Cin
        # arena-bound (VqaeTrainer): under capture a side branch of the graph, like the linear weight gradients
        with arena_fill():
            side = _wgrad_side_enter(x.device, (x, dy))
        if side is not None:
            ent, fork = side
            if not L.lib().wmz_conv2d_nhwc_wgrad_is_direct(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, dt):
                # a small layer: leaves with up to five others as ONE launch pair (graph nodes on the side branch cost host time)
                def queue():
                    _pending_conv.append(dict(x=x, dy=dy, gw=gw, gb=gb, need=need, ent=ent, fork=fork, dt=dt,
                                              geom=(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, 0, co, ci)))
                _defer(queue)
                return None, None

            def launch():
                ent[0].wait_event(fork)
                with torch.cuda.stream(ent[0]):
                    wss = _side_workspace(ent, x.device, need)
                    L.call('wmz_conv2d_nhwc_wgrad_ws', L.ptr(x), L.ptr(dy), L.ptr(gw), L.ptr(gb), B, Hi, Wi, Cin, Cout, KH, KW, stride,
                           pad, 0, co, ci, L.ptr(wss), wss.numel(), dt, L.stream())
            _defer(launch)
            return None, None
        ws = _workspace(x.device, need)
        L.call('wmz_conv2d_nhwc_wgrad_ws', L.ptr(x), L.ptr(dy), L.ptr(gw), L.ptr(gb), B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, 0,
               co, ci, L.ptr(ws), ws.numel(), dt, L.stream())
        return None, None
    ws = _workspace(x.device, need)
    dw = torch.empty((Cout, KH * KW * Cin), dtype=torch.float32, device=x.device)        # stored, not accumulated: no zero fill
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    L.call('wmz_conv2d_nhwc_wgrad_ws', L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, 1,
           0, 0, L.ptr(ws), ws.numel(), dt, L.stream())
    return dw, db


def bn_leaky_bwd_supported(x):
    """wmz_bn_leaky_bwd (the BatchNorm + LeakyReLU backward that recomputes its mask) is built for this tensor's channel count."""
    return bool(L.lib().wmz_bn_leaky_bwd_supported(x.shape[-1], L.dtype_code(x.dtype))) and x.is_contiguous() and x.data_ptr() % 16 == 0


def bn_act_bwd(x, y, dy, mean, rstd, gamma, leaky, slope=0.01, into=None, remask=None, add=None):
    """Training-mode BatchNorm (+ LeakyReLU) backward -> (dx, dgamma, dbeta, g) ; g = dy * act'(y).
    into = (dgamma, dbeta): fp32 [C] buffers that are ZERO on entry and receive the two column sums in place (the parameters'
    slots of a freshly zeroed gradient arena, each written once per step: no zero fills here, no `grad += g` afterwards).
    remask = (scale, shift) of the forward, for a LeakyReLU layer WITHOUT a skip input: the mask is recomputed from x, y is not
    read and g not produced (returned as None) -- wmz_bn_leaky_bwd, 5 tensor passes instead of 7.
    add (optional, like x): dx += add inside the apply pass -- the gradient x receives from its other consumer (a skip path)."""
    C = dy.shape[-1]
    M = dy.numel() // C
    dy = dy.contiguous()
    if into is not None:
        sgx, sg = into
        assert sg.dtype == torch.float32 and sgx.dtype == torch.float32 and sg.numel() == C and sgx.numel() == C
    else:
        sg, sgx = torch.zeros((2, C), dtype=torch.float32, device=dy.device).unbind(0)      # one fill for both
    dt = L.dtype_code(dy.dtype)
    dx = torch.empty_like(dy)
    if add is not None:
        add = add.contiguous()
        assert add.shape == dy.shape and add.dtype == dy.dtype
    if remask is not None:
        assert leaky and bn_leaky_bwd_supported(x) and dy.data_ptr() % 16 == 0
        L.call('wmz_bn_leaky_bwd', L.ptr(x), L.ptr(dy), L.ptr(remask[0]), L.ptr(remask[1]), L.ptr(mean), L.ptr(rstd), L.ptr(gamma),
               L.ptr(sg), L.ptr(sgx), L.ptr(add), L.ptr(dx), M, C, float(slope), dt, L.stream())
        return dx, sgx, sg, None
    g = torch.empty_like(dy)
    L.call('wmz_bn_act_bwd_reduce', L.ptr(x), L.ptr(y), L.ptr(dy), L.ptr(mean), L.ptr(rstd), L.ptr(g), L.ptr(sg),
           L.ptr(sgx), M, C, 1 if leaky else 0, float(slope), dt, L.stream())
    L.call('wmz_bn_bwd_apply_add', L.ptr(x), L.ptr(g), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(sg), L.ptr(sgx), L.ptr(add),
           L.ptr(dx), M, C, dt, L.stream())
    return dx, sgx, sg, g


def leaky_bwd(y, dy, slope=0.01):
    """g = dy * LeakyReLU'(y) from the stored output."""
    C = dy.shape[-1]
    M = dy.numel() // C
    dy = dy.contiguous()
    g = torch.empty_like(dy)
    L.call('wmz_bn_act_bwd_reduce', None, L.ptr(y), L.ptr(dy), None, None, L.ptr(g), None, None, M, C, 1, float(slope),
           L.dtype_code(dy.dtype), L.stream())
    return g


def bilinear2x_nhwc_bwd(dy):
    B, Ho, Wo, C = dy.shape
    dy = dy.contiguous()
    dx = torch.empty((B, Ho // 2, Wo // 2, C), dtype=dy.dtype, device=dy.device)
    L.call('wmz_bilinear2x_nhwc_bwd', L.ptr(dy), L.ptr(dx), B, Ho // 2, Wo // 2, C, L.dtype_code(dy.dtype), L.stream())
    return dx
