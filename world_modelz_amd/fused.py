"""Host side of wmz_layer_fused_fwd: weight / vector packing (cached per parameter version) and the fused
inference forward of Local3dAttentionTransformer (bf16, default widths)."""
import os
import torch

from . import _cast, config, ops
from . import _lib as L

D_, I_, M_ = 256, 128, 256
_PAD = 4 * 16384        # >= RING-1 slabs the kernel's LDS-DMA prefetch runs past the last real slab
MC_ = 32


def _pack_w(w):
    """[N, K] -> the kernel's stream of 1 KB MFMA 32x32x16 A operands, in (16-deep k-step, 32-feature block) order.
    Piece (s, b), lane l = 32*h + r, 8 elements: W[f(b, r)][h*K/2 + 8*s + j] -- the k axis is walked so that lane half h
    owns the contiguous features h*K/2.. of the activation, and output row r of block b is feature
    f = ho*N/2 + 16*b + i with (ho, i) the lane half / accumulator register that MFMA row r lands in
    (row = (i&3) + 8*(i>>2) + 4*ho), so every activation is a contiguous half row per lane."""
    N, K = w.shape
    NB, KS = N // 32, K // 16
    rho = torch.arange(32, device=w.device)
    ho, i = (rho >> 2) & 1, (rho & 3) + 4 * (rho >> 3)
    f = ho[None, :] * (16 * NB) + 16 * torch.arange(NB, device=w.device)[:, None] + i[None, :]      # [NB, 32]
    v = w[f].reshape(NB, 32, 2, KS, 8)                      # [b, r, h, s, j]
    return v.permute(3, 0, 2, 1, 4).reshape(-1)


def supported(transformer, x_dtype):
    """The fused per-token + row-attention kernels hold this stack: bf16 (speed mode) or float16 (the precise mode: inference
    only -- the callers that train pass the compute dtype, which is fp32 there), default widths, no dropout."""
    if x_dtype not in (torch.bfloat16, torch.float16) or len(transformer.layers) == 0:
        return False
    for attn, ff in transformer.layers:
        a, f = attn.fn, ff.fn
        if isinstance(a.to_out, torch.nn.Identity) or a.dropout > 0 or f.dropout > 0:
            return False
        if a.to_q.weight.shape != (I_, D_) or f.net[0].weight.shape != (M_, D_):
            return False
    return True


# ---------------------------------------------------------------------------------------------------------------------
# other widths: csrc/layer_chain.hip (16-token waves, MFMA 16x16x32, lane-group-major activations)

def chain_widths(transformer):
    """(D, I, M, MC) when every layer of `transformer` has the same widths and wmz_layer_chain_fwd_planes is built for them,
    else None."""
    import ctypes
    layers = list(transformer.layers)
    if not layers:
        return None
    a0, f0 = layers[0][0].fn, layers[0][1].fn
    if isinstance(a0.to_out, torch.nn.Identity) or not hasattr(a0, 'to_q'):
        return None
    I, D = a0.to_q.weight.shape
    M = f0.net[0].weight.shape[0]
    for attn, ff in layers:
        a, f = attn.fn, ff.fn
        if isinstance(a.to_out, torch.nn.Identity) or a.dropout > 0 or f.dropout > 0:
            return None
        if a.to_q.weight.shape != (I, D) or f.net[0].weight.shape != (M, D):
            return None
    mc = ctypes.c_int(0)
    if not L.lib().wmz_layer_chain_supported(D, I, M, ctypes.byref(mc)):
        return None
    return D, I, M, mc.value


def chain_supported(transformer, x_dtype, inference=False):
    # (training: bfloat16 only; the inference launch also has a half unit -- layer_chain_f16.hip, the precise mode)
    ok = (torch.bfloat16, torch.float16) if inference else (torch.bfloat16,)
    return x_dtype in ok and hasattr(transformer, 'pos_emb_s') and chain_widths(transformer) is not None


def chain_pays(widths, ntok, training):
    """Whether the chain kernels beat the op-by-op path at ntok tokens (bfloat16; config.chain_policy 'auto').  A chain workgroup
    holds 128 tokens and streams the layer's whole weight set through its LDS: 3 072 tokens are 24 workgroups on 256 CUs, each
    moving every weight byte, while the op-by-op GEMMs tile tokens AND features.  Measured crossovers
    (profiles/r06/time_chain_tokens.txt; weights per layer 0.2 / 0.33 / 1.2 / 2.6 MB at dim 96 / 128 / 384 / 512): forward ~2 000
    tokens per MB of weights; training step ~10 000 tokens (none below 0.25 MB)."""
    policy = config.get_chain_policy()
    if policy != 'auto':
        return policy == 'always'
    D, I, M = widths[:3]
    wbytes = 2 * (4 * D * I + 2 * D * M)
    if training:
        return wbytes < (1 << 18) or ntok >= 10240
    return ntok * 512 >= wbytes


def half_attention_ok(transformer, H, W):
    """The half attention unit (csrc/attn_fwd_row16_f16.hip) is built for these planes and heads (csrc/attn_fwd.hip)."""
    dh = {a.fn.to_q.weight.shape[0] // a.fn.heads for a, _ in transformer.layers}
    return dh <= {32, 64, 128} and (W == 16 or (W == 8 and H % 2 == 0))


def _chain_pieces(w, pad_value=0):
    """[N, K] fp32 -> the kernel's 1 KB MFMA 16x16x32 A operands in (k-step, 16-feature block) order, lane-linear:
    piece (ks, b), lane l = 16 ga + m, element j = W[(m >> 2) N/4 + 4 b + (m & 3)][ga K/4 + 8 ks + j] -- output feature and k
    axis both "lane-group-major" (csrc/layer_chain.hip), padded with zero pieces to whole slabs (pad_value: the filler -- an
    INDEX matrix is padded with -1)."""
    N, K = w.shape
    NB, KS = N // 16, K // 32
    dev = w.device
    m = torch.arange(16, device=dev)
    rows = (m >> 2)[None, :] * (N // 4) + 4 * torch.arange(NB, device=dev)[:, None] + (m & 3)[None, :]            # [NB, 16]
    ga = torch.arange(4, device=dev)
    cols = (ga[None, :, None] * (K // 4) + 8 * torch.arange(KS, device=dev)[:, None, None]
            + torch.arange(8, device=dev)[None, None, :])                                                             # [KS, 4, 8]
    t = w[rows.reshape(-1)][:, cols.reshape(-1)].reshape(NB, 16, KS, 4, 8).permute(2, 0, 3, 1, 4).reshape(-1, 512)   # [pieces, 512]
    pad = (-t.shape[0]) % L.lib().wmz_layer_chain_slab_pieces()
    if pad:
        t = torch.cat([t, t.new_full((pad, 512), pad_value)], 0)
    return t


def _chain_pack(head, tail, D, I, M, MC, dt=torch.bfloat16):
    """(wpack in dt -- bf16, or half for the _f16 unit --, vec fp32) for one launch of wmz_layer_chain_fwd_planes: stages to_out | MC-wide feed-forward chunks (W1'
    rows, then W2 columns) | q | k | v, LayerNorm affines folded in (W1' = W1 diag(g2), b1' = b1 + W1 be2; to_k / to_v likewise
    with the next layer's norm); vec = bout | b1' | b2 | bk' | bv'.  Cached per parameter version."""
    params = []
    if head is not None:
        attn, ff = head
        params += [attn.fn.to_out[0].weight, attn.fn.to_out[0].bias, ff.norm.weight, ff.norm.bias,
                   ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias]
    if tail is not None:
        an = tail[0]
        params += [an.norm.weight, an.norm.bias, an.fn.to_q.weight, an.fn.to_k.weight, an.fn.to_v.weight, an.fn.to_v.bias]

    def build(*ps):
        ps = [p.detach().float() for p in ps]
        dev = ps[0].device
        parts = []
        vec = torch.zeros(2 * D + M + 2 * I, dtype=torch.float32, device=dev)
        if head is not None:
            wout, bout, g2, be2, w1, b1, w2, b2 = ps[:8]
            parts.append(_chain_pieces(wout))
            w1f = w1 * g2[None, :]
            vec[:D] = bout
            vec[D:D + M] = b1 + w1 @ be2
            vec[D + M:2 * D + M] = b2
            for c in range(M // MC):
                parts.append(_chain_pieces(w1f[c * MC:(c + 1) * MC]))
                parts[-1] = parts[-1][:(MC // 16) * (D // 32)]                   # (no padding inside a chunk: its two GEMMs share slabs)
                parts.append(_chain_pieces(w2[:, c * MC:(c + 1) * MC])[:(D // 16) * (MC // 32)])
        if tail is not None:
            g1, be1, wq, wk, wv, bv = ps[-6:]
            parts.append(_chain_pieces(wq))
            parts.append(_chain_pieces(wk * g1[None, :]))
            parts.append(_chain_pieces(wv * g1[None, :]))
            vec[2 * D + M:2 * D + M + I] = wk @ be1
            vec[2 * D + M + I:] = bv + wv @ be1
        stream = torch.cat(parts, 0)
        sp = L.lib().wmz_layer_chain_slab_pieces()
        assert stream.shape[0] % sp == 0
        wpack = torch.cat([stream.reshape(-1), stream.new_zeros(3 * sp * 512)]).to(dt).contiguous()
        return wpack, vec
    return _cast.cached(params, f'chainpack{D}_{I}_{M}{_sfx(dt)}', build)


def transformer_forward_chain(tr, z):
    """Inference forward of Local3dAttentionTransformer on the chain kernel (bf16, widths of chain_widths): embedding launch,
    then per layer ONE attention launch + ONE per-token launch.  Returns the stream [B, S, H, W, D] (row-major)."""
    from . import functional as Fw
    D, I, M, MC = chain_widths(tr)
    layers = list(tr.layers)
    B, S, H, W = z.shape
    HW = H * W
    dev, bf = z.device, config.get_fused_dtype()       # (bf: the stream's 16-bit format -- bfloat16, or half in the precise mode)
    x = Fw.embed_tokens(z, tr.embedding.weight, tr.pos_emb_s.weight, tr.pos_emb_h.weight, tr.pos_emb_w.weight)
    if x.dtype != bf:
        x = x.to(bf)                                   # (precise mode: the embedding sum in fp32, rounded once to half)
    assert x.is_contiguous()

    def launch(o, x_in, head, tail):
        wpack, vec = _chain_pack(head, tail, D, I, M, MC, bf)
        xo = torch.empty((B, S, H, W, D), dtype=bf, device=dev) if head is not None else None
        q = torch.empty((B, S, H, W, I), dtype=bf, device=dev) if tail is not None else None
        kv = torch.empty((2, B, S, H, W, I), dtype=bf, device=dev) if tail is not None else None
        L.call('wmz_layer_chain_fwd_planes' + _sfx(bf), L.ptr(o), L.ptr(x_in), L.ptr(xo), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec),
               B, S, S, HW, D, I, M, 1 if head is not None else 0, 1 if tail is not None else 0, 1e-5, L.stream())
        return xo, q, kv
    _, q, kv = launch(None, x, None, layers[0])
    for l, (attn, ff) in enumerate(layers):
        heads, ext = attn.fn.heads, attn.fn.extents
        o, _, _ = ops.local3d_attention_fwd(q, kv[0], kv[1], ext, heads)
        x, q, kv = launch(o, x, (attn, ff), layers[l + 1] if l + 1 < len(layers) else None)
    return x


class ChainLayoutError(RuntimeError):
    """ChainPackSet cannot describe this model's parameter layout (the trainer then takes the op-by-op path)."""


class ChainPackSet:
    """The training step's weight streams of csrc/layer_chain.hip / layer_chain_bwd.hip for EVERY launch of the stack -- forward
    launches, feed-forward-side and attention-side backward launches -- rebuilt from the flat parameter arena by a handful of
    launches: a packed stream is a fixed permutation of the raw parameters times (where a LayerNorm weight is folded in) one
    gamma element each, so index tables are built ONCE from the inference packer's own piece order (`_chain_pieces` on index
    matrices) and refresh() = two gathers, a product and a cast; the folded bias terms W beta are one batched matrix-vector
    product per kind over strided views of the arena.  Buffer addresses never change (the captured training graph replays on
    them)."""

    def __init__(self, tr, arena):
        self.widths = D, I, M, MC = chain_widths(tr)
        layers = list(tr.layers)
        nl = len(layers)
        dev = arena.flat_param.device
        off = {id(p): o for p, o in zip(arena.params, arena.offsets)}

        def idx(p):
            return off[id(p)] + torch.arange(p.numel(), device=dev, dtype=torch.int64).view(p.shape)

        def cols(vec_p, like):                           # index matrix of gamma[k] broadcast over the rows of a [N, K] weight
            return idx(vec_p)[None, :].expand(like.shape[0], -1)
        sp = L.lib().wmz_layer_chain_slab_pieces()
        one = torch.full((1, 1), -1, dtype=torch.int64, device=dev)                   # scale index -1 = no scale

        def pieces(w_idx, s_idx=None, n=None):
            """(source indices, scale indices) of the pieces of an index matrix, optionally only its first n pieces."""
            a = _chain_pieces(w_idx, -1)
            b = _chain_pieces(s_idx, -1) if s_idx is not None else torch.full_like(a, -1)
            return (a, b) if n is None else (a[:n], b[:n])
        streams, self.slices = [], {}
        pos = 0

        def add(key, parts):
            nonlocal pos
            w = torch.cat([p[0] for p in parts], 0).reshape(-1)
            sc = torch.cat([p[1] for p in parts], 0).reshape(-1)
            assert (w.numel() // 512) % sp == 0
            streams.append((w, sc))
            self.slices[key] = pos
            pos += w.numel()
        nvec = 2 * D + M + 2 * I
        vidx = torch.full((nl + 1, nvec), -1, dtype=torch.int64, device=dev)
        for l in range(nl + 1):
            head = layers[l - 1] if l > 0 else None
            tail = layers[l] if l < nl else None
            parts = []
            if head is not None:
                attn, ff = head
                w1, w2 = idx(ff.fn.net[0].weight), idx(ff.fn.net[3].weight)
                g1 = cols(ff.norm.weight, w1)
                parts.append(pieces(idx(attn.fn.to_out[0].weight)))
                for c in range(M // MC):
                    parts.append(pieces(w1[c * MC:(c + 1) * MC], g1[c * MC:(c + 1) * MC], (MC // 16) * (D // 32)))
                    parts.append(pieces(w2[:, c * MC:(c + 1) * MC], None, (D // 16) * (MC // 32)))
                vidx[l, :D] = idx(attn.fn.to_out[0].bias)
                vidx[l, D:D + M] = idx(ff.fn.net[0].bias)
                vidx[l, D + M:2 * D + M] = idx(ff.fn.net[3].bias)
            if tail is not None:
                an = tail[0]
                wk, wv = idx(an.fn.to_k.weight), idx(an.fn.to_v.weight)
                parts += [pieces(idx(an.fn.to_q.weight)), pieces(wk, cols(an.norm.weight, wk)), pieces(wv, cols(an.norm.weight, wv))]
                vidx[l, 2 * D + M + I:] = idx(an.fn.to_v.bias)
            add(('fwd', l), parts)
        for l, (attn, ff) in enumerate(layers):
            # backward streams: the TRANSPOSED weights, the norms' weights folded in on the side that faces the normalised rows
            w1, w2 = idx(ff.fn.net[0].weight), idx(ff.fn.net[3].weight)
            g1 = cols(ff.norm.weight, w1)
            parts = []
            for c in range(M // MC):
                parts.append(pieces(w2[:, c * MC:(c + 1) * MC].t(), None, (MC // 16) * (D // 32)))
                parts.append(pieces(w1[c * MC:(c + 1) * MC].t(), g1[c * MC:(c + 1) * MC].t(), (D // 16) * (MC // 32)))
            parts.append(pieces(idx(attn.fn.to_out[0].weight).t()))
            add(('ff_bwd', l), parts)
            wk, wv = idx(attn.fn.to_k.weight), idx(attn.fn.to_v.weight)
            ga = cols(attn.norm.weight, wk)
            add(('qkv_bwd', l), [pieces(wk.t(), ga.t()), pieces(wv.t(), ga.t()), pieces(idx(attn.fn.to_q.weight).t())])
        tail_pad = torch.full((3 * sp * 512,), -1, dtype=torch.int64, device=dev)     # the prefetch runs past the last slab
        widx = torch.cat([w for w, _ in streams] + [tail_pad])
        sidx = torch.cat([sc for _, sc in streams] + [tail_pad])
        self.flat = arena.flat_param
        assert arena.flat_param.numel() < 2 ** 31 and widx.numel() < 2 ** 31
        # tables (round 5): 32-bit indices, a byte mask for the padding, and the LayerNorm-weight factors only for the
        # elements that HAVE one (the w1 / k / v pieces: about half) as a compact (position, source) pair -- 31 -> ~15 bytes per
        # packed element (0.7 -> 0.34 GB at dim 384 / depth 20) and half the bytes a refresh moves
        self.widx, self.wpad = widx.clamp(min=0).to(torch.int32), (widx < 0)
        self.spos = torch.nonzero(sidx >= 0).reshape(-1)                     # (int64: index_copy_ takes nothing narrower)
        self.ssrc = sidx[sidx >= 0].to(torch.int32)
        self.w32 = torch.empty(widx.numel(), dtype=torch.float32, device=dev)
        self.sw = torch.empty(self.spos.numel(), dtype=torch.float32, device=dev)
        self.sg = torch.empty(self.spos.numel(), dtype=torch.float32, device=dev)
        self.wpack = torch.empty(widx.numel(), dtype=torch.bfloat16, device=dev)
        self.vidx, self.vpad = vidx.reshape(-1).clamp(min=0).to(torch.int32), (vidx.reshape(-1) < 0)
        self.vec = torch.empty((nl + 1, nvec), dtype=torch.float32, device=dev)
        # the folded bias terms: b1' = b1 + W1 beta_ff, bk' = Wk beta_attn, bv' = bv + Wv beta_attn, batched over the layers (a
        # layer's parameters sit at one stride in the arena: strided [L, N, K] / [L, K, 1] views, one bmm per kind)
        def strided(ps, shape):
            o = [off[id(p)] for p in ps]
            st = o[1] - o[0] if len(o) > 1 else 0
            if any(o[i + 1] - o[i] != st for i in range(len(o) - 1)):
                return None
            n = ps[0].numel()
            inner = (shape[1], 1) if len(shape) == 2 else (1, 1)
            return torch.as_strided(self.flat, (len(ps),) + tuple(shape), (st,) + inner, o[0])
        W1 = strided([ff.fn.net[0].weight for _, ff in layers], (M, D))
        bff = strided([ff.norm.bias for _, ff in layers], (D, 1))
        Wk = strided([a.fn.to_k.weight for a, _ in layers], (I, D))
        Wv = strided([a.fn.to_v.weight for a, _ in layers], (I, D))
        bat = strided([a.norm.bias for a, _ in layers], (D, 1))
        if any(t is None for t in (W1, bff, Wk, Wv, bat)):
            # (frozen or re-ordered layers: no strided view of the arena holds them -- the caller trains op by op instead)
            raise ChainLayoutError('the transformer layers are not laid out at one stride in the parameter arena')
        self._mv = [(W1, bff, self.vec[1:, D:D + M]), (Wk, bat, self.vec[:nl, 2 * D + M:2 * D + M + I]),
                    (Wv, bat, self.vec[:nl, 2 * D + M + I:])]
        self.refresh()

    def refresh(self):
        flat = self.flat.detach()
        torch.index_select(flat, 0, self.widx, out=self.w32)
        self.w32.masked_fill_(self.wpad, 0.0)
        torch.index_select(self.w32, 0, self.spos, out=self.sw)          # the elements a LayerNorm weight is folded into
        torch.index_select(flat, 0, self.ssrc, out=self.sg)
        self.w32.index_copy_(0, self.spos, self.sw.mul_(self.sg))
        self.wpack.copy_(self.w32)
        torch.index_select(flat, 0, self.vidx, out=self.vec.view(-1))
        self.vec.view(-1).masked_fill_(self.vpad, 0.0)
        for W, beta, dst in self._mv:
            dst.add_(torch.bmm(W.detach(), beta.detach()).squeeze(-1))

    def stream(self, kind, l):
        """The packed stream of launch (kind, l): 'fwd' l = 0 .. depth (0: the embedding's tail-only launch), 'ff_bwd' / 'qkv_bwd'
        l = layer.  What lies behind a stream in the buffer is its readable padding."""
        return self.wpack[self.slices[(kind, l)]:]


def _chain_layer_train(packs, l, o, x_in, head, tail, keep_raw):
    """Launch l of the training forward on the chain kernel -> dict of everything it wrote."""
    D, I, M, MC = packs.widths
    ntok = x_in.numel() // D
    dev, bf = x_in.device, torch.bfloat16
    r = {}
    if head:
        names = (('x', D), ('xh_ff', D), ('z', M), ('h', M)) + ((('x1', D),) if keep_raw else ())
        for k, w in names:
            r[k] = torch.empty((ntok, w), dtype=bf, device=dev)
        r['st_ff'] = torch.empty((2, ntok), dtype=torch.float32, device=dev)
    if tail:
        r['q'] = torch.empty((ntok, I), dtype=bf, device=dev)
        r['kv'] = torch.empty((ntok, 2 * I), dtype=bf, device=dev)
        r['xh_attn'] = torch.empty((ntok, D), dtype=bf, device=dev)
        r['st_attn'] = torch.empty((2, ntok), dtype=torch.float32, device=dev)
    g = r.get
    L.call('wmz_layer_chain_fwd_train', L.ptr(o), L.ptr(x_in), L.ptr(g('x')), L.ptr(g('q')), L.ptr(g('kv')),
           L.ptr(packs.stream('fwd', l)), L.ptr(packs.vec[l]), L.ptr(g('x1')), L.ptr(g('xh_ff')), L.ptr(g('z')), L.ptr(g('h')),
           L.ptr(g('st_ff')), L.ptr(g('xh_attn')), L.ptr(g('st_attn')), ntok, D, I, M, 1 if head else 0, 1 if tail else 0, 1e-5,
           L.stream())
    return r


def _chain_layer_backward(packs, l, attn, ff, dy, x_in, xh_attn, st_attn, q, kv, o, lse, xh_ff, st_ff, z, h):
    """One layer of the stack's backward on the chain kernels: wmz_chain_ff_bwd -> attention backward -> wmz_chain_qkv_bwd, the
    weight gradients as plain GEMMs over the operands the forward and these kernels wrote (normalised rows behind the norms), the
    LayerNorm affine gradients from the raw weight gradients (wmz_ln_affine_grads) -- the structure of _layer_backward_fused."""
    D, I, M, MC = packs.widths
    an_g, an_b, wq, wk, wv, bv, wout, bout, fn_g, fn_b, w1, b1, w2, b2 = _layer_params(attn, ff)
    dev, bf = dy.device, torch.bfloat16
    ntok = dy.numel() // D
    dy2 = dy.reshape(ntok, D)
    dz = torch.empty((ntok, M), dtype=bf, device=dev)
    dx1 = torch.empty((ntok, D), dtype=bf, device=dev)
    do = torch.empty(o.shape, dtype=bf, device=dev)
    L.call('wmz_chain_ff_bwd', L.ptr(dy2), L.ptr(z), L.ptr(xh_ff), L.ptr(st_ff[1]), L.ptr(dz), L.ptr(dx1), L.ptr(do),
           L.ptr(packs.stream('ff_bwd', l)), ntok, D, I, M, L.stream())
    dq, dkv = ops.local3d_attention_bwd(q, kv[..., :I], kv[..., I:], o, lse, do, attn.fn.extents, attn.fn.heads)
    dx = torch.empty(dy.shape, dtype=bf, device=dev)
    L.call('wmz_chain_qkv_bwd', L.ptr(dq), L.ptr(dkv), L.ptr(xh_attn), L.ptr(st_attn[1]), L.ptr(dx1), L.ptr(dx),
           L.ptr(packs.stream('qkv_bwd', l)), ntok, D, I, L.stream())
    s_ff2, s_out, s_q = _GradSink(w2, b2), _GradSink(wout, bout), _GradSink(wq)
    G1 = torch.empty((M, D), dtype=torch.float32, device=dev)
    c1 = torch.empty((M,), dtype=torch.float32, device=dev)
    Gkv = torch.empty((2 * I, D), dtype=torch.float32, device=dev)
    ckv = torch.empty((2 * I,), dtype=torch.float32, device=dev)
    # ONE launch pair: at dim 384 all five fill >= 3/8 of their 256-wide tiles (wgrad3_kernel's bar; one ineligible problem would
    # send the whole batch to the 128-wide kernel: 225 us a layer)
    wide = [(dy2, h, s_ff2.bufs[0], s_ff2.bufs[1], False),                     # dW2 = dy^T GELU(z), db2 = colsum(dy)
            (dz, xh_ff, G1, c1, True),                                          # against the NORMALISED rows: raw gradient + column sums
            (dkv.reshape(ntok, 2 * I), xh_attn, Gkv, ckv, True)]
    narrow = [(dx1, o.reshape(ntok, I), s_out.bufs[0], s_out.bufs[1], False),
              (dq.reshape(ntok, I), x_in.reshape(ntok, D), s_q.bufs[0], None, False)]
    s_ff1 = _GradSink(w1, b1, fn_g, fn_b)
    s_kv = _GradSink(wk, wv, bv, an_g, an_b)
    bk_, bw_ = s_kv.bufs[0], s_kv.bufs[1]
    adjacent = (wk.is_contiguous() and wv.is_contiguous() and bk_.is_contiguous() and bw_.is_contiguous()
                and wv.data_ptr() == wk.data_ptr() + 4 * wk.numel() and bw_.data_ptr() == bk_.data_ptr() + 4 * bk_.numel())
    probs = [(G1, c1, w1, fn_g, fn_b, s_ff1.bufs[0], s_ff1.bufs[1], s_ff1.bufs[2], s_ff1.bufs[3], M, D, 0)]
    if adjacent:
        probs.append((Gkv, ckv, wk, an_g, an_b, bk_, s_kv.bufs[2], s_kv.bufs[3], s_kv.bufs[4], 2 * I, D, I))
    else:
        probs.append((Gkv[:I], ckv[:I], wk, an_g, an_b, bk_, None, s_kv.bufs[3], s_kv.bufs[4], I, D, I))
        probs.append((Gkv[I:], ckv[I:], wv, an_g, an_b, bw_, s_kv.bufs[2], s_kv.bufs[3], s_kv.bufs[4], I, D, 0))

    def weight_grads(side):
        ops.linear_wgrad_batch(wide + narrow, side=side)
        _ln_affine_grads_batch(probs)
    # nothing in the backward chain reads these results: under capture they leave on the weight-gradient side branch (gradients that
    # land in the flat arena only: a gradient handed back to autograd is consumed on the compute stream)
    direct = all(sk.direct for sk in (s_ff2, s_out, s_q, s_ff1, s_kv))
    if direct and config.get_wgrad_stream():
        ops.side_branch(dev, (dy2, h, dz, xh_ff, dkv, xh_attn, dx1, o, dq, x_in, G1, c1, Gkv, ckv), weight_grads)
    else:
        weight_grads(None)
    g_wk, g_wv, g_bv, g_ag, g_ab = s_kv.done()
    (g_wq,) = s_q.done()
    g_wout, g_bout = s_out.done()
    g_w1, g_b1, g_fg, g_fb = s_ff1.done()
    g_w2, g_b2 = s_ff2.done()
    return dx, [g_ag, g_ab, g_wq, g_wk, g_wv, g_bv, g_wout, g_bout, g_fg, g_fb, g_w1, g_b1, g_w2, g_b2]


class _ChainTrainForward(torch.autograd.Function):
    """The whole stack as one autograd node for the widths of csrc/layer_chain.hip (the reference's published runs): forward =
    embedding + per layer ONE attention launch + ONE per-token launch that also leaves what the backward reads (normalised rows,
    pre-activation, GELU of it, LayerNorm statistics); backward per layer = wmz_chain_ff_bwd + the attention backward +
    wmz_chain_qkv_bwd + one batched weight-gradient launch pair (config.fused_backward() off: the op-by-op block backward
    functions instead, reading the raw rows the forward then keeps as well)."""

    @staticmethod
    def forward(ctx, tr, packs, z, last_only, *params):
        from . import functional as Fw
        D, I, M, MC = packs.widths
        layers = list(tr.layers)
        B, S, H, W = z.shape
        fused_bwd = config.fused_backward()
        x0 = Fw.embed_tokens(z, tr.embedding.weight.detach(), tr.pos_emb_s.weight.detach(), tr.pos_emb_h.weight.detach(),
                             tr.pos_emb_w.weight.detach())
        cur = _chain_layer_train(packs, 0, None, x0, False, True, False)
        x_in = x0.reshape(-1, D)
        saved = []
        empty = x0.new_empty(0)
        for l, (attn, ff) in enumerate(layers):
            q, kv = cur['q'].view(B, S, H, W, I), cur['kv'].view(B, S, H, W, 2 * I)
            o, lse, _ = ops.local3d_attention_fwd(q, kv[..., :I], kv[..., I:], attn.fn.extents, attn.fn.heads, need_lse=True)
            nxt = _chain_layer_train(packs, l + 1, o, x_in, True, l + 1 < len(layers), not fused_bwd)
            saved += [x_in, cur['xh_attn'], cur['st_attn'], q, kv, o, lse, nxt.get('x1', empty), nxt['xh_ff'], nxt['st_ff'],
                      nxt['z'], nxt['h']]
            x_in, cur = nxt['x'], nxt
        ctx.tr, ctx.packs, ctx.last_only, ctx.fused_bwd = tr, packs, bool(last_only), fused_bwd
        ctx.save_for_backward(z, *saved)
        xo = x_in.view(B, S, H, W, D)
        return xo[:, -1].contiguous() if last_only else xo

    @staticmethod
    def backward(ctx, dy):
        from . import backward as Bk
        tr, packs = ctx.tr, ctx.packs
        D, I, M, MC = packs.widths
        layers = list(tr.layers)
        z, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        B, S, H, W = z.shape
        grads = [None] * (14 * len(layers))
        dy = dy.contiguous()
        if ctx.last_only:
            full = torch.zeros((B, S, H, W, D), dtype=dy.dtype, device=dy.device)
            full[:, -1] = dy
            dy = full
        NS = 12
        for l in range(len(layers) - 1, -1, -1):
            attn, ff = layers[l]
            x_in, xh_attn, st_attn, q, kv, o, lse, x1, xh_ff, st_ff, zpre, hact = saved[NS * l:NS * l + NS]
            if ctx.fused_bwd:
                dy, g = _chain_layer_backward(packs, l, attn, ff, dy, x_in, xh_attn, st_attn, q, kv, o, lse, xh_ff, st_ff, zpre, hact)
                grads[14 * l:14 * l + 14] = g
                continue
            an_g, an_b, wq, wk, wv, bv, wout, bout, fn_g, fn_b, w1, b1, w2, b2 = _layer_params(attn, ff)
            x1v, x_inv = x1.view(B, S, H, W, D), x_in.view(B, S, H, W, D)
            cf = _Ctx((x1v, fn_g, fn_b, w1, b1, w2, b2, zpre.view(B, S, H, W, M), hact.view(B, S, H, W, M)),
                      has_res=True, res_is_x=True, ln_stats=(st_ff[0], st_ff[1]))
            dx1, g_fg, g_fb, g_w1, g_b1, g_w2, g_b2 = Bk.feed_forward_block_backward(cf, dy)[:7]
            ca = _Ctx((x_inv, x_inv, an_g, an_b, wq, wk, wv, bv, wout, bout, q, kv, o, lse), extents=attn.fn.extents,
                      heads=attn.fn.heads, has_res=True, res_is_xkv=True, same_src=True, ln_stats=(st_attn[0], st_attn[1]))
            r = Bk.attention_block_backward(ca, dx1)
            dy = r[0]
            grads[14 * l:14 * l + 14] = [r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], g_fg, g_fb, g_w1, g_b1, g_w2, g_b2]
        ce = _Ctx((z,), params=(tr.embedding.weight, tr.pos_emb_s.weight, tr.pos_emb_h.weight, tr.pos_emb_w.weight))
        ge = Bk.embed_backward(ce, dy)
        return (None, None, None, None, ge[1], ge[2], ge[3], ge[4], *grads)


def transformer_forward_chain_train(tr, packs, z, last_only=False):
    """Training forward of the stack on the chain kernel (ChainPackSet of the trainer's arena)."""
    params = [tr.embedding.weight, tr.pos_emb_s.weight, tr.pos_emb_h.weight, tr.pos_emb_w.weight]
    for attn, ff in tr.layers:
        params += _layer_params(attn, ff)
    return _ChainTrainForward.apply(tr, packs, z, last_only, *params)


def _sfx(dt):
    return '_f16' if dt == torch.float16 else ''


def _layer_pack(head, tail, dt=torch.bfloat16):
    """head / tail: (attn PreNorm, ff PreNorm) of the layer whose to_out+FF run, and of the layer whose q|k|v run.
    Returns (wpack bf16, vec fp32) built by ONE launch of wmz_layer_fused_pack from the fp32 parameters: the weights in
    the kernel's streaming order (see _pack_w for the element order; W1 rows one chunk ahead of the W2 columns:
    W1[0], W1[1], W2[0], W1[2], W2[1], .., W1[7], W2[6], W2[7]) with the LayerNorm affines folded in (W1' = W1 diag(g2),
    b1' = b1 + W1 be2; same for to_k / to_v with the next layer's norm), and bout | b1' | b2 | bk' | bv'."""
    params = []
    if head is not None:
        attn, ff = head
        params += [attn.fn.to_out[0].weight, attn.fn.to_out[0].bias, ff.norm.weight, ff.norm.bias,
                   ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias]
    if tail is not None:
        an = tail[0]
        params += [an.norm.weight, an.norm.bias, an.fn.to_q.weight, an.fn.to_k.weight, an.fn.to_v.weight, an.fn.to_v.bias]

    def build(*ps):
        ps = [p.detach() for p in ps]
        assert all(p.dtype == torch.float32 and p.is_contiguous() for p in ps)
        hp = ps[:8] if head is not None else [None] * 8
        tp = ps[-6:] if tail is not None else [None] * 6
        nw = (D_ * I_ + 2 * M_ * D_ if head is not None else 0) + (3 * I_ * D_ if tail is not None else 0)
        dev = ps[0].device
        wpack = torch.empty(nw + _PAD // 2, dtype=dt, device=dev)
        vec = torch.empty(2048, dtype=torch.float32, device=dev)
        L.call('wmz_layer_fused_pack' + _sfx(dt), *[L.ptr(t) for t in hp], *[L.ptr(t) for t in tp], L.ptr(wpack), L.ptr(vec),
               D_, I_, M_, L.stream())
        return wpack, vec
    return _cast.cached(params, 'fusedpack' + _sfx(dt), build)


def _layer_pack_bwd(attn, ff):
    """(wpack_qkv, wpack_ff): the TRANSPOSED weight streams of one layer for wmz_qkv_fused_bwd / wmz_ff_fused_bwd, built by
    wmz_layer_fused_bwd_pack from the fp32 parameters (cached per parameter version like the forward streams)."""
    a, f = attn.fn, ff.fn
    params = [a.to_q.weight, a.to_k.weight, a.to_v.weight, attn.norm.weight, a.to_out[0].weight, f.net[0].weight,
              ff.norm.weight, f.net[3].weight]

    def build(*ps):
        ps = [p.detach() for p in ps]
        assert all(p.dtype == torch.float32 and p.is_contiguous() for p in ps)
        dev = ps[0].device
        wq = torch.empty(3 * D_ * I_ + _PAD // 2, dtype=torch.bfloat16, device=dev)
        wf = torch.empty(2 * M_ * D_ + D_ * I_ + _PAD // 2, dtype=torch.bfloat16, device=dev)
        L.call('wmz_layer_fused_bwd_pack', *[L.ptr(t) for t in ps], L.ptr(wq), L.ptr(wf), D_, I_, M_, L.stream())
        return wq, wf
    return _cast.cached(params, 'fusedpackbwd', build)


class PackSet:
    """Every packed weight stream / vector block the fused kernels of one transformer read -- the forward boundaries
    (_layer_pack) and each layer's two backward streams (_layer_pack_bwd) -- rebuilt together by ONE call of
    wmz_fused_pack_table (two launches) instead of a launch pair per stream.  The block descriptors are device-resident
    tables built once (rebuilt if a parameter moves); refresh() fills the persistent buffers and stamps the very cache
    entries _layer_pack / _layer_pack_bwd look up, so the forward and backward code is unchanged.  Training calls it after
    every optimizer step (train.py), next to _cast.BulkOperands.refresh()."""

    def __init__(self, tr, backward=True):
        self.tr = tr
        self.backward = backward
        self._ptrs = None
        self._build()

    def _params(self):
        return [p for p in self.tr.parameters()]

    def _build(self):
        import numpy as np
        layers = list(self.tr.layers)
        dev = layers[0][0].norm.weight.device
        rows, jobs, self.entries = [], [], []
        state = {'g8': 0}

        def ptr(t, off=0):
            return 0 if t is None else t.data_ptr() + 4 * off

        def block(dst, doff, w, woff, rs, ks, N, K, gn, gk, gamma, rgamma):
            rows.append([ptr(w, woff), rs, ks, N, K, gn, gk, ptr(gamma), ptr(rgamma), dst.data_ptr() + 2 * doff, state['g8']])
            state['g8'] += N * K // 8
            return doff + N * K

        bounds = [(None, layers[0])] + [(layers[l], layers[l + 1] if l + 1 < len(layers) else None) for l in range(len(layers))]
        for head, tail in bounds:
            nw = (D_ * I_ + 2 * M_ * D_ if head is not None else 0) + (3 * I_ * D_ if tail is not None else 0)
            wpack = torch.zeros(nw + _PAD // 2, dtype=torch.bfloat16, device=dev)
            vec = torch.zeros(2048, dtype=torch.float32, device=dev)
            params, off = [], 0
            job = [0] * 10
            if head is not None:
                attn, ff = head
                wout, bout, g2, be2 = attn.fn.to_out[0].weight, attn.fn.to_out[0].bias, ff.norm.weight, ff.norm.bias
                w1, b1, w2, b2 = ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias
                params += [wout, bout, g2, be2, w1, b1, w2, b2]
                off = block(wpack, off, wout, 0, I_, 1, D_, I_, D_, I_, None, None)
                off = block(wpack, off, w1, 0, D_, 1, MC_, D_, MC_, D_, g2, None)
                for c in range(1, M_ // MC_):
                    off = block(wpack, off, w1, c * MC_ * D_, D_, 1, MC_, D_, MC_, D_, g2, None)
                    off = block(wpack, off, w2, (c - 1) * MC_, M_, 1, D_, MC_, D_, MC_, None, None)
                off = block(wpack, off, w2, (M_ // MC_ - 1) * MC_, M_, 1, D_, MC_, D_, MC_, None, None)
                job[0:5] = [ptr(bout), ptr(b1), ptr(w1), ptr(be2), ptr(b2)]
            if tail is not None:
                an = tail[0]
                g1, be1 = an.norm.weight, an.norm.bias
                wq, wk, wv, bv = an.fn.to_q.weight, an.fn.to_k.weight, an.fn.to_v.weight, an.fn.to_v.bias
                params += [g1, be1, wq, wk, wv, bv]
                off = block(wpack, off, wq, 0, D_, 1, I_, D_, I_, D_, None, None)
                off = block(wpack, off, wk, 0, D_, 1, I_, D_, I_, D_, g1, None)
                off = block(wpack, off, wv, 0, D_, 1, I_, D_, I_, D_, g1, None)
                job[5:9] = [ptr(wk), ptr(wv), ptr(be1), ptr(bv)]
            assert off == nw
            job[9] = vec.data_ptr()
            jobs.append(job)
            self.entries.append((tuple(params), 'fusedpack', (wpack, vec)))
        if self.backward:
            for attn, ff in layers:
                a, f = attn.fn, ff.fn
                wq, wk, wv, g1 = a.to_q.weight, a.to_k.weight, a.to_v.weight, attn.norm.weight
                wout, w1, g2, w2 = a.to_out[0].weight, f.net[0].weight, ff.norm.weight, f.net[3].weight
                sq = torch.zeros(3 * D_ * I_ + _PAD // 2, dtype=torch.bfloat16, device=dev)
                sf = torch.zeros(2 * M_ * D_ + D_ * I_ + _PAD // 2, dtype=torch.bfloat16, device=dev)
                off = block(sq, 0, wk, 0, 1, D_, D_, I_, 128, 128, None, g1)
                off = block(sq, off, wv, 0, 1, D_, D_, I_, 128, 128, None, g1)
                off = block(sq, off, wq, 0, 1, D_, D_, I_, 128, 128, None, None)
                off = 0
                for c in range(M_ // 32):
                    off = block(sf, off, w2, c * 32, 1, M_, 32, D_, 32, 128, None, None)
                off = block(sf, off, w1, 0, 1, D_, D_, M_, 128, 32, None, g2)
                off = block(sf, off, wout, 0, 1, I_, I_, D_, 128, 128, None, None)
                self.entries.append(((wq, wk, wv, g1, wout, w1, g2, w2), 'fusedpackbwd', (sq, sf)))
        self.nblk = len(rows)
        self.total8 = state['g8']
        rows.append([0] * 10 + [self.total8])
        self.rows = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
        self.jobs = torch.from_numpy(np.asarray(jobs, dtype=np.int64)).to(dev)
        self.njobs = len(jobs)
        self._ptrs = [p.data_ptr() for p in self._params()]

    def refresh(self):
        import weakref
        ps = self._params()
        assert all(p.dtype == torch.float32 and p.is_contiguous() for p in ps)
        if [p.data_ptr() for p in ps] != self._ptrs:
            self._build()                                  # a parameter moved (.to(), re-allocation): new tables
        L.call('wmz_fused_pack_table', L.ptr(self.rows), self.nblk, self.total8, L.ptr(self.jobs), self.njobs, D_, I_, M_,
               L.stream())
        for params, tag, val in self.entries:
            ver = (_cast._epoch,) + tuple((p._version, p.data_ptr()) for p in params)
            _cast._cache[_cast._key(params, None, tag)] = (ver, val, tuple(weakref.ref(p) for p in params))


X_IN_TILED, X_OUT_TILED, X1_NORMALISED, XRM_NORMALISED = 1, 2, 4, 8      # include/wmz.h WMZ_FUSED_X*


def layer_fused(o, x, head, tail, eps=1e-5, xflags=0):
    """One launch of the fused per-token kernel on row-major (or, with xflags, tiled-stream) x.
    Returns (x_out | None, q | None, kv | None)."""
    ntok = x.numel() // D_
    wpack, vec = _layer_pack(head, tail, x.dtype)
    lead = x.shape[:-1]
    xo = torch.empty_like(x) if head is not None else None
    q = torch.empty(lead + (I_,), dtype=x.dtype, device=x.device) if tail is not None else None
    kv = torch.empty((2,) + lead + (I_,), dtype=x.dtype, device=x.device) if tail is not None else None
    L.call('wmz_layer_fused_fwd_planes' + _sfx(x.dtype), L.ptr(o), L.ptr(x), L.ptr(xo), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec),
           1, 1, 1, ntok, D_, I_, M_, 1 if head is not None else 0, 1 if tail is not None else 0, int(xflags), float(eps),
           L.stream())
    return xo, q, kv


def cone_planes(S, eS, depth):
    """Planes each layer has to produce so that the LAST plane of the last layer is exact: need[l] = planes of queries
    at layer l, src[l] = planes of that layer's input stream (= its key/value planes).  main.py:37 reads x[:, -1] only and
    a query at plane s sees planes s-eS..s+eS (local_3d_attention.py:95-104), so need[L-1] = 1, need[l] = need[l+1] + eS,
    capped at S."""
    need = [0] * depth
    need[depth - 1] = 1
    for l in range(depth - 2, -1, -1):
        need[l] = min(S, need[l + 1] + eS)
    src = [min(S, need[l] + eS) for l in range(depth)]
    return need, src


_side_streams = {}


def _run(tr, z, cone):
    """embedding -> depth x [attention launch, per-token launch].  cone: only the planes the last frame depends on.
    With config.clip_streams > 1 the clips are cut into that many groups, each group's chain on its own stream."""
    from . import config
    B = z.shape[0]
    layers = list(tr.layers)
    eS = int(layers[0][0].fn.extents[0])
    use_cone = cone and all(int(a.fn.extents[0]) == eS for a, _ in layers)
    planes = cone_planes(z.shape[1], eS, len(layers))[1][0] if use_cone else z.shape[1]
    groups = min(config.get_clip_streams(), B)
    # (a chain needs at least half a chip of workgroups -- one per plane -- to be worth its own stream: the dependence cone of
    #  a small batch does not)
    while groups > 1 and (B // groups) * planes < 128:
        groups -= 1
    if groups <= 1:
        return _run_chain(tr, z, cone, None)
    n_out = (cone_planes(z.shape[1], eS, len(layers))[0][-1]) if use_cone else z.shape[1]
    dt = config.get_fused_dtype()
    out = torch.empty((B, n_out) + tuple(z.shape[2:]) + (D_,), dtype=dt, device=z.device)
    cur = torch.cuda.current_stream()
    pool = _side_streams.setdefault(z.device, [])
    while len(pool) < groups - 1:
        pool.append(torch.cuda.Stream(device=z.device))
    z = z.contiguous()
    bounds = [(g * B) // groups for g in range(groups + 1)]
    # the weight streams are packed (cached per parameter version) on the caller's stream BEFORE the fork: every chain reads them
    for l in range(len(layers)):
        _layer_pack(None if l == 0 else layers[l - 1], layers[l], dt)
    _layer_pack(layers[-1], None, dt)
    for g in range(1, groups):
        st = pool[g - 1]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            _run_chain(tr, z[bounds[g]:bounds[g + 1]], cone, out[bounds[g]:bounds[g + 1]])
    _run_chain(tr, z[bounds[0]:bounds[1]], cone, out[bounds[0]:bounds[1]])
    for g in range(1, groups):
        cur.wait_stream(pool[g - 1])
    return out


def _run_chain(tr, z, cone, out):
    """One chain of launches over the clips z on the current stream; the last launch writes into `out` (a batch slice of the
    caller's result) when given."""
    layers = list(tr.layers)
    depth = len(layers)
    B, S, H, W = z.shape
    HW = H * W
    eS = int(layers[0][0].fn.extents[0])
    if cone and any(int(a.fn.extents[0]) != eS for a, _ in layers):
        cone = False
    need, src = cone_planes(S, eS, depth) if cone else ([S] * depth, [S] * depth)
    tiled = HW % 32 == 0                      # whole 32-token tiles per plane: the stream between layers stays tiled
    from . import config
    dev, bf = z.device, config.get_fused_dtype()          # (bf: bfloat16, or float16 in the precise mode)
    sfx = _sfx(bf)
    n0 = src[0]
    wpack, vec = _layer_pack(None, layers[0], bf)
    x = torch.empty((B, n0, H, W, D_), dtype=bf, device=dev)
    q = torch.empty((B, n0, H, W, I_), dtype=bf, device=dev)
    kv = torch.empty((2, B, n0, H, W, I_), dtype=bf, device=dev)          # k planes, then v planes
    L.call('wmz_embed_qkv_fused_fwd_planes' + sfx, L.ptr(z.contiguous()), L.ptr(tr.embedding.weight.detach()),
           L.ptr(tr.pos_emb_s.weight.detach()), L.ptr(tr.pos_emb_h.weight.detach()), L.ptr(tr.pos_emb_w.weight.detach()),
           L.ptr(x), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec), B, S, H, W, n0, D_, I_, M_,
           tr.embedding.num_embeddings, X_OUT_TILED if tiled else 0, 1e-5, L.stream())
    for l, (attn, ff) in enumerate(layers):
        n_in, n_q = src[l], need[l]
        heads, ext = attn.fn.heads, attn.fn.extents
        o = torch.empty((B, n_q, H, W, I_), dtype=bf, device=dev)
        if ops._profile_hook is not None:
            ops._profile_hook('wmz_local3d_attn_fwd', True)
        L.call('wmz_local3d_attn_fwd_planes', L.ptr(q), L.ptr(kv[0]), L.ptr(kv[1]), L.ptr(o), None,
               B, n_in, H, W, heads, I_ // heads, int(ext[0]), int(ext[1]), int(ext[2]), I_, I_, I_, I_,
               n_in - n_q, n_q, L.dtype_code(bf), L.stream())
        if ops._profile_hook is not None:
            ops._profile_hook('wmz_local3d_attn_fwd', False)
        tail = layers[l + 1] if l + 1 < depth else None
        wpack, vec = _layer_pack((attn, ff), tail, bf)
        xo = out if (tail is None and out is not None) else torch.empty((B, n_q, H, W, D_), dtype=bf, device=dev)
        q = torch.empty((B, n_q, H, W, I_), dtype=bf, device=dev) if tail is not None else None
        kv = torch.empty((2, B, n_q, H, W, I_), dtype=bf, device=dev) if tail is not None else None
        xflags = (X_IN_TILED if tiled else 0) | (X_OUT_TILED if tiled and tail is not None else 0)
        L.call('wmz_layer_fused_fwd_planes' + sfx, L.ptr(o), L.ptr(x), L.ptr(xo), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec),
               B, n_q, n_in, HW, D_, I_, M_, 1, 1 if tail is not None else 0, xflags, 1e-5, L.stream())
        x = xo
    return x


def transformer_forward(tr, x=None, z=None):
    """depth x [attention, feed-forward] on the fused kernels: per layer ONE attention launch + ONE per-token launch, the
    embedding riding in the first per-token launch.  Returns the stream [B, S, H, W, D] (row-major)."""
    if z is not None:
        return _run(tr, z, cone=False)
    layers = list(tr.layers)
    _, q, kv = layer_fused(None, x, None, layers[0])
    for l, (attn, ff) in enumerate(layers):
        o, _, _ = ops.local3d_attention_fwd(q, kv[0], kv[1], attn.fn.extents, attn.fn.heads)
        x, q, kv = layer_fused(o, x, (attn, ff), layers[l + 1] if l + 1 < len(layers) else None)
    return x


def transformer_forward_last(tr, z):
    """The last plane of transformer_forward(tr, z=z) ([B, H, W, D]) computing only its dependence cone: identical
    arithmetic per token (bit-identical result), the planes that cannot reach the last frame are never launched."""
    return _run(tr, z, cone=True)[:, 0]


# ---------------------------------------------------------------------------------------------------------------------
# training forward on the fused kernels (wmz_*_train): one attention launch + one per-token launch per layer, and the
# tensors the op-by-op backward (backward.py) reads are written row-major on the way

def _embed_train(tr, z, tiled, xhat_rm=False):
    """xhat_rm (tiled only): the row-major output holds the NORMALISED rows (what the fused backward reads); the raw stream
    lives in the tiled output."""
    B, S, H, W = z.shape
    dev, bf = z.device, torch.bfloat16
    wpack, vec = _layer_pack(None, tr.layers[0])
    x_rm = torch.empty((B, S, H, W, D_), dtype=bf, device=dev)
    x_t = torch.empty((B, S, H, W, D_), dtype=bf, device=dev) if tiled else None
    q = torch.empty((B, S, H, W, I_), dtype=bf, device=dev)
    kv = torch.empty((B, S, H, W, 2 * I_), dtype=bf, device=dev)
    st_attn = torch.empty((2, B * S * H * W), dtype=torch.float32, device=dev)       # LayerNorm statistics for the backward
    L.call('wmz_embed_qkv_fused_fwd_train', L.ptr(z.contiguous()), L.ptr(tr.embedding.weight.detach()),
           L.ptr(tr.pos_emb_s.weight.detach()), L.ptr(tr.pos_emb_h.weight.detach()), L.ptr(tr.pos_emb_w.weight.detach()),
           L.ptr(x_t if tiled else x_rm), L.ptr(x_rm if tiled else None), L.ptr(q), L.ptr(kv), L.ptr(st_attn), L.ptr(wpack),
           L.ptr(vec), B, S, H, W, D_, I_, M_, tr.embedding.num_embeddings,
           (X_OUT_TILED if tiled else 0) | (XRM_NORMALISED if (tiled and xhat_rm) else 0), 1e-5, L.stream())
    return (x_t if tiled else x_rm), x_rm, q, kv, st_attn


def _layer_train(o, x_in, head, tail, tiled, save_z, xhat_rm=False):
    """x_in: the stream in the layout the previous launch left it in (tiled if `tiled`).  Returns (x_next, x_rm, x1, q, kv,
    st_ff, st_attn, zt): x_next in that same layout for the next launch (None after the last layer), x_rm / x1 row-major for
    the backward, st_* the [2, ntok] LayerNorm statistics (feed-forward's norm; the next layer's attention norm or None), zt
    the feed-forward pre-activation in the fused backward's tiled layout (None unless save_z)."""
    lead = o.shape[:-1]
    dev, bf = o.device, torch.bfloat16
    ntok = o.numel() // I_
    wpack, vec = _layer_pack(head, tail)
    x_rm = torch.empty(lead + (D_,), dtype=bf, device=dev)
    x1 = torch.empty(lead + (D_,), dtype=bf, device=dev)
    out_tiled = tiled and tail is not None
    x_t = torch.empty(lead + (D_,), dtype=bf, device=dev) if out_tiled else None
    q = torch.empty(lead + (I_,), dtype=bf, device=dev) if tail is not None else None
    kv = torch.empty(lead + (2 * I_,), dtype=bf, device=dev) if tail is not None else None
    # the fused backward (save_z) needs x1 only as its NORMALISED rows (LayerNorm backward, dW1 operand): the kernel has them in
    # registers as the W1 operand and stores those instead of x1
    xflags = (X_IN_TILED if tiled else 0) | (X_OUT_TILED if out_tiled else 0) | (X1_NORMALISED if save_z else 0)
    if xhat_rm and out_tiled:          # x_rm = the next layer's NORMALISED input rows; its raw input is x_t
        xflags |= XRM_NORMALISED
    st_ff = torch.empty((2, ntok), dtype=torch.float32, device=dev)
    st_attn = torch.empty((2, ntok), dtype=torch.float32, device=dev) if tail is not None else None
    zt = torch.empty((ntok, M_), dtype=bf, device=dev) if save_z else None
    L.call('wmz_layer_fused_fwd_train', L.ptr(o), L.ptr(x_in), L.ptr(x_t if out_tiled else x_rm),
           L.ptr(x_rm if out_tiled else None), L.ptr(x1), L.ptr(q), L.ptr(kv), L.ptr(st_ff), L.ptr(st_attn), L.ptr(zt),
           L.ptr(wpack), L.ptr(vec), ntok, D_, I_, M_, 1, 1 if tail is not None else 0, xflags, 1e-5, L.stream())
    return (x_t if out_tiled else (x_rm if tail is not None else None)), x_rm, x1, q, kv, st_ff, st_attn, zt


def _layer_params(attn, ff):
    a, f = attn.fn, ff.fn
    return [attn.norm.weight, attn.norm.bias, a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_v.bias,
            a.to_out[0].weight, a.to_out[0].bias, ff.norm.weight, ff.norm.bias,
            f.net[0].weight, f.net[0].bias, f.net[3].weight, f.net[3].bias]


class _GradSink:
    """Where a parameter's gradient goes: its slice of the flat gradient arena (parallel.FlatArena: the kernels accumulate
    straight into it and the data-parallel reducer is told when the layer's last gradient has landed) or a fresh zero
    tensor that is handed back to autograd."""

    def __init__(self, *params):
        self.params = params
        self.bufs = [getattr(p, '_wmz_grad', None) for p in params]
        self.direct = all(b is not None for b in self.bufs)
        if not self.direct:
            self.bufs = [torch.zeros(p.shape, dtype=torch.float32, device=p.device) for p in params]

    def done(self):
        """-> the tensors autograd receives (None in arena mode)."""
        if self.direct:
            for p in self.params:
                ready = getattr(p, '_wmz_ready', None)
                if ready is not None:
                    ready()
            return [None] * len(self.params)
        return self.bufs


_zero_rows = {}


def _zero_row(dev):
    z = _zero_rows.get(dev)
    if z is None:
        z = _zero_rows[dev] = torch.zeros(D_, dtype=torch.bfloat16, device=dev)
    return z


def _ln_affine_grads_batch(probs):
    """wmz_ln_affine_grads_batch on a list of (G, s, W, gamma, beta, dW, dbias | None, dgamma, dbeta, N, K, bias_from)."""
    import ctypes
    n = len(probs)
    vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
    cols = [vp() for _ in range(9)]
    Ns, Ks, bf = ci(), ci(), ci()
    for i, pr in enumerate(probs):
        for j in range(9):
            t = pr[j]
            cols[j][i] = L.ptr(t.detach() if (t is not None and t.requires_grad) else t)
        Ns[i], Ks[i], bf[i] = pr[9], pr[10], pr[11]
    L.call('wmz_ln_affine_grads_batch', n, *cols, Ns, Ks, bf, L.stream())


def _layer_backward_fused(attn, ff, dy, x_in, q, kv, o, lse, x1, st_attn, st_ff, zt, dy_last=None, x_in_tiled=None):
    """One layer of the stack's backward on the fused per-token kernels: wmz_ff_fused_bwd -> attention backward ->
    wmz_qkv_fused_bwd, the weight gradients as plain GEMMs over the operands those kernels write, the LayerNorm affine
    gradients from the raw weight gradients (wmz_ln_affine_grads).  x1: the NORMALISED rows of the feed-forward block's input, as
    the forward stored them (WMZ_FUSED_X1_NORMALISED).  x_in_tiled given: x_in holds the layer input's NORMALISED rows too
    (WMZ_FUSED_XRM_NORMALISED) and x_in_tiled the raw stream in the tiled layout.  dy_last = (S, HW): dy holds only the clips' last planes
    ([B, H, W, D], the last layer under the denoiser's last-frame loss).  Returns (gradient w.r.t. the layer's input, the 14
    parameter gradients in _layer_params order)."""
    an_g, an_b, wq, wk, wv, bv, wout, bout, fn_g, fn_b, w1, b1, w2, b2 = _layer_params(attn, ff)
    dev, bf = dy.device, torch.bfloat16
    lead = x1.shape[:-1]
    ntok = x1.numel() // D_
    wpack_qkv, wpack_ff = _layer_pack_bwd(attn, ff)
    g = torch.empty((ntok, M_), dtype=bf, device=dev)
    dz = torch.empty((ntok, M_), dtype=bf, device=dev)
    dx1 = torch.empty((ntok, D_), dtype=bf, device=dev)
    do = torch.empty(lead + (I_,), dtype=bf, device=dev)
    if dy_last is None:
        L.call('wmz_ff_fused_bwd', L.ptr(dy), L.ptr(zt), L.ptr(x1), L.ptr(st_ff), L.ptr(g), L.ptr(dz), None, L.ptr(dx1),
               L.ptr(do), L.ptr(wpack_ff), ntok, D_, I_, M_, 0, 0, None, L.stream())
        dy2, g2 = dy.reshape(ntok, D_), g
    else:
        S_, HW_ = dy_last
        L.call('wmz_ff_fused_bwd', L.ptr(dy), L.ptr(zt), L.ptr(x1), L.ptr(st_ff), L.ptr(g), L.ptr(dz), None, L.ptr(dx1),
               L.ptr(do), L.ptr(wpack_ff), ntok, D_, I_, M_, S_, HW_, L.ptr(_zero_row(dev)), L.stream())
        # dW2 = dy^T GELU(z) only has the last planes' rows to sum over
        dy2 = dy.reshape(-1, D_)
        g2 = g.view(-1, S_, HW_, M_)[:, -1].reshape(-1, M_)
    # ---- attention core
    dq, dkv = ops.local3d_attention_bwd(q, kv[..., :I_], kv[..., I_:], o, lse, do, attn.fn.extents, attn.fn.heads)
    # ---- to_q / to_k / to_v inputs
    dx = torch.empty(lead + (D_,), dtype=bf, device=dev)
    if x_in_tiled is None:
        xhat = torch.empty((ntok, D_), dtype=bf, device=dev)
        x_q, x_q_tiled = x_in.reshape(ntok, D_), False
    else:
        xhat = None
        x_q, x_q_tiled = x_in_tiled, True
    L.call('wmz_qkv_fused_bwd', L.ptr(dq), I_, L.ptr(dkv), 2 * I_, L.ptr(x_in), L.ptr(st_attn), L.ptr(dx1), L.ptr(dx),
           L.ptr(xhat), L.ptr(wpack_qkv), ntok, D_, I_, L.stream())
    if xhat is None:
        xhat = x_in.reshape(ntok, D_)
    # ---- the layer's five weight gradients: plain GEMMs over the token axis, ONE launch pair.  Those behind a LayerNorm
    # are taken against the NORMALISED input (raw gradients G, column sums c) and turned into parameter gradients below.
    s_ff2, s_out, s_q = _GradSink(w2, b2), _GradSink(wout, bout), _GradSink(wq)
    G1 = torch.empty((M_, D_), dtype=torch.float32, device=dev)
    c1 = torch.empty((M_,), dtype=torch.float32, device=dev)
    Gkv = torch.empty((2 * I_, D_), dtype=torch.float32, device=dev)
    ckv = torch.empty((2 * I_,), dtype=torch.float32, device=dev)
    ops.linear_wgrad_batch([
        (dy2, g2, s_ff2.bufs[0], s_ff2.bufs[1], False),                       # dW2 = dy^T GELU(z), db2 = colsum(dy)
        (dz, x1.reshape(ntok, D_), G1, c1, True),            # x1 = the NORMALISED rows the forward stored
        (dx1, o.reshape(ntok, I_), s_out.bufs[0], s_out.bufs[1], False),
        (dq.reshape(ntok, I_), x_q, s_q.bufs[0], None, False, x_q_tiled),
        (dkv.reshape(ntok, 2 * I_), xhat, Gkv, ckv, True)])
    s_ff1 = _GradSink(w1, b1, fn_g, fn_b)
    s_kv = _GradSink(wk, wv, bv, an_g, an_b)
    bk_, bw_ = s_kv.bufs[0], s_kv.bufs[1]
    adjacent = (wk.is_contiguous() and wv.is_contiguous() and bk_.is_contiguous() and bw_.is_contiguous()
                and wv.data_ptr() == wk.data_ptr() + 4 * wk.numel() and bw_.data_ptr() == bk_.data_ptr() + 4 * bk_.numel())
    # one launch for the layer's LayerNorm-affine conversions: (G, s, W, gamma, beta, dW, dbias, dgamma, dbeta, N, K, bias_from)
    probs = [(G1, c1, w1, fn_g, fn_b, s_ff1.bufs[0], s_ff1.bufs[1], s_ff1.bufs[2], s_ff1.bufs[3], M_, D_, 0)]
    # (the kernel only does pointer arithmetic on W and dW: two separately allocated neighbours are as good as one arena)
    if adjacent:                        # FlatArena: to_k.weight | to_v.weight (and their gradients) are one [2I, D] block
        probs.append((Gkv, ckv, wk, an_g, an_b, bk_, s_kv.bufs[2], s_kv.bufs[3], s_kv.bufs[4], 2 * I_, D_, I_))
    else:
        probs.append((Gkv[:I_], ckv[:I_], wk, an_g, an_b, bk_, None, s_kv.bufs[3], s_kv.bufs[4], I_, D_, I_))
        probs.append((Gkv[I_:], ckv[I_:], wv, an_g, an_b, bw_, s_kv.bufs[2], s_kv.bufs[3], s_kv.bufs[4], I_, D_, 0))
    _ln_affine_grads_batch(probs)
    g_wk, g_wv, g_bv, g_ag, g_ab = s_kv.done()
    (g_wq,) = s_q.done()
    g_wout, g_bout = s_out.done()
    g_w1, g_b1, g_fg, g_fb = s_ff1.done()
    g_w2, g_b2 = s_ff2.done()
    return dx, [g_ag, g_ab, g_wq, g_wk, g_wv, g_bv, g_wout, g_bout, g_fg, g_fb, g_w1, g_b1, g_w2, g_b2]


class _Ctx:                      # what backward.attention_block_backward / feed_forward_block_backward read off a ctx
    def __init__(self, saved, **kw):
        self.saved_tensors = saved
        self.__dict__.update(kw)


class _TrainForward(torch.autograd.Function):
    """The whole transformer stack (embedding + depth x [attention, feed-forward]) as one autograd node: forward on the
    fused kernels, backward layer by layer through the same block backward functions as the op-by-op path (the
    feed-forward pre-activation is recomputed there by one LayerNorm-GEMM instead of being stored)."""

    @staticmethod
    def forward(ctx, tr, z, last_only, *params):
        layers = list(tr.layers)
        B, S, H, W = z.shape
        tiled = (H * W) % 32 == 0
        # the fused backward kernels (layer_fused_bwd.hip) work on whole 32-token tiles and read the pre-activation the
        # forward leaves behind; otherwise the op-by-op backward recomputes it
        fused_bwd = config.fused_backward() and (B * S * H * W) % 32 == 0
        # fused backward on a tiled stream: a layer's row-major input copy holds the NORMALISED rows (all the backward needs
        # of them: LayerNorm backward, to_k | to_v weight gradient); the raw rows -- the to_q weight gradient's operand -- are
        # read from the tiled stream the forward kernels hand each other
        xhat_rm = fused_bwd and tiled
        x_cur, x_rm, q, kv, st_attn = _embed_train(tr, z, tiled, xhat_rm)
        saved = []
        for l, (attn, ff) in enumerate(layers):
            o, lse, _ = ops.local3d_attention_fwd(q, kv[..., :I_], kv[..., I_:], attn.fn.extents, attn.fn.heads, need_lse=True)
            x_in_rm, x_in_t = x_rm, (x_cur if xhat_rm else lse.new_empty(0))
            x_cur, x_rm, x1, q_n, kv_n, st_ff, st_attn_n, zt = _layer_train(o, x_cur, (attn, ff),
                                                                           layers[l + 1] if l + 1 < len(layers) else None,
                                                                           tiled, fused_bwd, xhat_rm)
            saved += [x_in_rm, q, kv, o, lse, x1, st_attn, st_ff, zt if zt is not None else lse.new_empty(0), x_in_t]
            q, kv, st_attn = q_n, kv_n, st_attn_n
        ctx.tr = tr
        ctx.fused_bwd = fused_bwd
        ctx.last_only = bool(last_only)
        ctx.save_for_backward(z, *saved)
        # last_only: the caller reads the last plane only (main.py:37) and hands back a gradient for it alone
        return x_rm[:, -1].contiguous() if last_only else x_rm

    @staticmethod
    def backward(ctx, dy):
        from . import backward as Bk
        from .functional import LN_EPS
        tr = ctx.tr
        layers = list(tr.layers)
        z, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        grads = [None] * (14 * len(layers))
        dy = dy.contiguous()
        NS = 10
        dy_last = None
        if ctx.last_only:
            B, S, H, W = z.shape
            if ctx.fused_bwd:
                dy_last = (S, H * W)                       # the last layer's kernel reads the other planes' zeros from a zero row
            else:
                full = torch.zeros((B, S, H, W, D_), dtype=dy.dtype, device=dy.device)
                full[:, -1] = dy
                dy = full
        for l in range(len(layers) - 1, -1, -1):
            attn, ff = layers[l]
            x_in, q, kv, o, lse, x1, st_attn, st_ff, zt, x_in_t = saved[NS * l:NS * l + NS]
            an_g, an_b, wq, wk, wv, bv, wout, bout, fn_g, fn_b, w1, b1, w2, b2 = _layer_params(attn, ff)
            if ctx.fused_bwd:
                dy, g = _layer_backward_fused(attn, ff, dy, x_in, q, kv, o, lse, x1, st_attn, st_ff, zt,
                                              dy_last if l == len(layers) - 1 else None, x_in_t if x_in_t.numel() else None)
                grads[14 * l:14 * l + 14] = g
                continue
            dt = x1.dtype
            # feed-forward block: y = W2 GELU(W1 LN(x1) + b1) + b2 + x1
            stats = (st_ff[0], st_ff[1])                   # computed by the fused forward: no extra pass over x1
            zpre, hact = ops.linear_fwd_gelu_pair(x1, _cast.operand(w1, dt), bias=b1.detach(), ln=(fn_g.detach(), fn_b.detach()),
                                                  ln_eps=LN_EPS, ln_stats=stats)
            cf = _Ctx((x1, fn_g, fn_b, w1, b1, w2, b2, zpre, hact), has_res=True, res_is_x=True, ln_stats=stats)
            dx1, g_fg, g_fb, g_w1, g_b1, g_w2, g_b2 = Bk.feed_forward_block_backward(cf, dy)[:7]
            # attention block: x1 = to_out(attn(LN(x), q = x)) + x
            ca = _Ctx((x_in, x_in, an_g, an_b, wq, wk, wv, bv, wout, bout, q, kv, o, lse), extents=attn.fn.extents,
                      heads=attn.fn.heads, has_res=True, res_is_xkv=True, same_src=True, ln_stats=(st_attn[0], st_attn[1]))
            r = Bk.attention_block_backward(ca, dx1)
            dy = r[0]
            grads[14 * l:14 * l + 14] = [r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], g_fg, g_fb, g_w1, g_b1, g_w2, g_b2]
        ce = _Ctx((z,), params=(tr.embedding.weight, tr.pos_emb_s.weight, tr.pos_emb_h.weight, tr.pos_emb_w.weight))
        ge = Bk.embed_backward(ce, dy)
        return (None, None, None, ge[1], ge[2], ge[3], ge[4], *grads)


def transformer_forward_train(tr, z, last_only=False):
    """The stack's output [B, S, H, W, D] -- or, with last_only, its last plane [B, H, W, D] (what the denoiser's loss reads:
    the backward then never materialises the zero gradient of the other planes)."""
    params = [tr.embedding.weight, tr.pos_emb_s.weight, tr.pos_emb_h.weight, tr.pos_emb_w.weight]
    for attn, ff in tr.layers:
        params += _layer_params(attn, ff)
    return _TrainForward.apply(tr, z, last_only, *params)
