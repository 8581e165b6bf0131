"""Host side of wmz_layer_fused_fwd: weight / vector packing (cached per parameter version) and the fused
inference forward of Local3dAttentionTransformer (bf16, default widths)."""
import torch

from . import _cast, ops
from . import _lib as L

D_, I_, M_ = 256, 128, 256
_PAD = 4 * 16384        # >= RING-1 slabs the kernel's LDS-DMA prefetch runs past the last real slab
MC_ = 64


def _pack_w(w):
    """[N, K] -> [K/32][N][4 chunks][8]: the MFMA 16x16x32 A-operand rows of one 32-deep k-step, with the 16-byte chunk
    index of row n XOR-ed by (-(n>>2)) & 3 (conflict-free ds_read_b128 in the kernel)."""
    N, K = w.shape
    v = w.reshape(N, K // 32, 4, 8)
    n = torch.arange(N, device=w.device)
    f = (-(n >> 2)) & 3                                       # [N]
    src = torch.arange(4, device=w.device)[None, :] ^ f[:, None]   # physical chunk c' holds logical chunk c'^f
    v = torch.gather(v, 2, src[:, None, :, None].expand(N, K // 32, 4, 8))
    return v.permute(1, 0, 2, 3).reshape(-1)


def supported(transformer, x_dtype):
    if x_dtype != torch.bfloat16 or len(transformer.layers) == 0:
        return False
    for attn, ff in transformer.layers:
        a, f = attn.fn, ff.fn
        if isinstance(a.to_out, torch.nn.Identity) or a.dropout > 0 or f.dropout > 0:
            return False
        if a.to_q.weight.shape != (I_, D_) or f.net[0].weight.shape != (M_, D_):
            return False
    return True


def _layer_pack(head, tail):
    """head / tail: (attn PreNorm, ff PreNorm) of the layer whose to_out+FF run, and of the layer whose q|k|v run."""
    params, parts = [], []
    if head is not None:
        attn, ff = head
        params += [attn.fn.to_out[0].weight, ff.fn.net[0].weight, ff.fn.net[3].weight]
    if tail is not None:
        attn_n = tail[0]
        params += [attn_n.fn.to_q.weight, attn_n.fn.to_k.weight, attn_n.fn.to_v.weight]

    def build_w(*ws):
        ws = [w.detach().to(torch.bfloat16) for w in ws]
        out, i = [], 0
        if head is not None:
            out.append(_pack_w(ws[0]))
            for c in range(M_ // MC_):          # feed-forward streamed MC hidden units at a time: W1 rows, then W2 columns
                out += [_pack_w(ws[1][c * MC_:(c + 1) * MC_]), _pack_w(ws[2][:, c * MC_:(c + 1) * MC_])]
            i = 3
        if tail is not None:
            out += [_pack_w(ws[i]), _pack_w(ws[i + 1]), _pack_w(ws[i + 2])]
        out.append(torch.zeros(_PAD // 2, dtype=torch.bfloat16, device=ws[0].device))
        return torch.cat(out)
    wpack = _cast.operand(tuple(params), torch.bfloat16, 'fusedw', build_w)

    vparams = []
    if head is not None:
        attn, ff = head
        vparams += [attn.fn.to_out[0].bias, ff.norm.weight, ff.norm.bias, ff.fn.net[0].bias, ff.fn.net[3].bias]
    if tail is not None:
        attn_n = tail[0]
        vparams += [attn_n.norm.weight, attn_n.norm.bias, attn_n.fn.to_v.bias]

    def build_v(*vs):
        dev = vs[0].device
        z = lambda n: torch.zeros(n, device=dev)  # noqa: E731
        vs = [v.detach().float() for v in vs]
        if head is not None:
            hv, rest = vs[:5], vs[5:]
        else:
            hv, rest = [z(D_), z(D_), z(D_), z(M_), z(D_)], vs
        tv = [rest[0], rest[1], torch.cat([z(I_), rest[2]])] if tail is not None else [z(D_), z(D_), z(2 * I_)]
        return torch.cat(hv + tv)
    vec = _cast.operand(tuple(vparams), torch.float32, 'fusedv', build_v)
    return wpack, vec


def layer_fused(o, x, head, tail, eps=1e-5):
    """Returns (x_out | None, q | None, kv | None)."""
    ntok = x.numel() // D_
    wpack, vec = _layer_pack(head, tail)
    lead = x.shape[:-1]
    xo = torch.empty_like(x) if head is not None else None
    q = torch.empty(lead + (I_,), dtype=x.dtype, device=x.device) if tail is not None else None
    kv = torch.empty(lead + (2 * I_,), dtype=x.dtype, device=x.device) if tail is not None else None
    L.call('wmz_layer_fused_fwd', L.ptr(o), L.ptr(x), L.ptr(xo), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec), ntok,
           D_, I_, M_, 1 if head is not None else 0, 1 if tail is not None else 0, float(eps), L.stream())
    return xo, q, kv


def embed_qkv_fused(tr, z, eps=1e-5):
    """Embedding + layer 0's q | k|v in one launch.  Returns (x, q, kv)."""
    B, S, H, W = z.shape
    wpack, vec = _layer_pack(None, tr.layers[0])
    dev = z.device
    x = torch.empty((B, S, H, W, D_), dtype=torch.bfloat16, device=dev)
    q = torch.empty((B, S, H, W, I_), dtype=torch.bfloat16, device=dev)
    kv = torch.empty((B, S, H, W, 2 * I_), dtype=torch.bfloat16, device=dev)
    L.call('wmz_embed_qkv_fused_fwd', L.ptr(z.contiguous()), L.ptr(tr.embedding.weight.detach()),
           L.ptr(tr.pos_emb_s.weight.detach()), L.ptr(tr.pos_emb_h.weight.detach()), L.ptr(tr.pos_emb_w.weight.detach()),
           L.ptr(x), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec), B, S, H, W, D_, I_, M_, tr.embedding.num_embeddings,
           float(eps), L.stream())
    return x, q, kv


def transformer_forward(tr, x=None, z=None):
    """depth x [attention, feed-forward] on the fused kernels: per layer ONE attention launch + ONE per-token launch;
    with `z` (token grid) the embedding rides in the first per-token launch."""
    layers = list(tr.layers)
    if z is not None:
        x, q, kv = embed_qkv_fused(tr, z)
    else:
        _, q, kv = layer_fused(None, x, None, layers[0])
    for l, (attn, ff) in enumerate(layers):
        o, _, _ = ops.local3d_attention_fwd(q, kv[..., :I_], kv[..., I_:], attn.fn.extents, attn.fn.heads)
        x, q, kv = layer_fused(o, x, (attn, ff), layers[l + 1] if l + 1 < len(layers) else None)
    return x


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


def transformer_forward_last(tr, z):
    """The last plane of transformer_forward(tr, z=z) ([B, H, W, D]) computing only its dependence cone: identical
    arithmetic per token (bit-identical result), the planes that cannot reach the last frame are never launched."""
    layers = list(tr.layers)
    depth = len(layers)
    B, S, H, W = z.shape
    HW = H * W
    eS = int(layers[0][0].fn.extents[0])
    if any(int(a.fn.extents[0]) != eS for a, _ in layers):
        return transformer_forward(tr, z=z)[:, -1]
    need, src = cone_planes(S, eS, depth)
    dev = z.device
    bf = torch.bfloat16
    n0 = src[0]
    wpack, vec = _layer_pack(None, layers[0])
    x = torch.empty((B, n0, H, W, D_), dtype=bf, device=dev)
    q = torch.empty((B, n0, H, W, I_), dtype=bf, device=dev)
    kv = torch.empty((B, n0, H, W, 2 * I_), dtype=bf, device=dev)
    L.call('wmz_embed_qkv_fused_fwd_planes', L.ptr(z.contiguous()), L.ptr(tr.embedding.weight.detach()),
           L.ptr(tr.pos_emb_s.weight.detach()), L.ptr(tr.pos_emb_h.weight.detach()), L.ptr(tr.pos_emb_w.weight.detach()),
           L.ptr(x), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec), B, S, H, W, n0, D_, I_, M_,
           tr.embedding.num_embeddings, 1e-5, L.stream())
    for l, (attn, ff) in enumerate(layers):
        n_in, n_q = src[l], need[l]
        heads = attn.fn.heads
        ext = attn.fn.extents
        o = torch.empty((B, n_q, H, W, I_), dtype=bf, device=dev)
        L.call('wmz_local3d_attn_fwd_planes', L.ptr(q), L.ptr(kv), L.ptr(kv[..., I_:]), L.ptr(o), None,
               B, n_in, H, W, heads, I_ // heads, int(ext[0]), int(ext[1]), int(ext[2]), I_, 2 * I_, 2 * I_, I_,
               n_in - n_q, n_q, L.dtype_code(bf), L.stream())
        tail = layers[l + 1] if l + 1 < depth else None
        wpack, vec = _layer_pack((attn, ff), tail)
        xo = torch.empty((B, n_q, H, W, D_), dtype=bf, device=dev)
        q = torch.empty((B, n_q, H, W, I_), dtype=bf, device=dev) if tail is not None else None
        kv = torch.empty((B, n_q, H, W, 2 * I_), dtype=bf, device=dev) if tail is not None else None
        L.call('wmz_layer_fused_fwd_planes', L.ptr(o), L.ptr(x), L.ptr(xo), L.ptr(q), L.ptr(kv), L.ptr(wpack), L.ptr(vec),
               B, n_q, n_in, HW, D_, I_, M_, 1, 1 if tail is not None else 0, 1e-5, L.stream())
        x = xo
    return x[:, 0]
