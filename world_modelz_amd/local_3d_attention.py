"""MI355X drop-in for vq-video-diffusion/local_3d_attention.py.

Same classes, constructor signatures, attribute names and state_dict keys as the reference
(PreNorm :11-17, FeedForward :20-31, Local3dAttention :34-118, Local3dAttentionTransformer :121-163), so
reference checkpoints load and `from local_3d_attention import Local3dAttentionTransformer` keeps working
(see world_modelz_amd/dropin/).  The bodies call the fused HIP blocks in functional.py; nothing here runs on
the CPU and nothing materialises the unfolded key/value windows.

Dtype rule at the module boundary: every public `forward()` returns its result in the dtype of its floating-point
input (token-index inputs: the dtype of the module's parameters), whatever the compute dtype (config.py) is -- so
reference code that composes these modules with plain torch layers (main.py:33-36 feeds the transformer's output to an
fp32 nn.Linear) keeps working in the bf16 speed mode.  The stack's own layer loop and the drop-in
VqVideoDiffusionModel use the internal entry points (`forward_prenorm`, `forward_compute`) and stay in the compute
dtype end to end.
"""
import torch
from torch import nn

from . import functional as Fw


class PreNorm(nn.Module):
    """LayerNorm in front of `fn` (reference :11-17).  Only the positional input is normalised; keyword
    arguments (the attention's `q`) pass through untouched -- quirk Q1."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        fused = getattr(self.fn, 'forward_prenorm', None)
        if fused is not None:
            return fused(x, self.norm, **kwargs).to(x.dtype)         # LayerNorm rides in the GEMM prologue
        return self.fn(nn.functional.layer_norm(x, self.norm.normalized_shape, self.norm.weight, self.norm.bias,
                                                self.norm.eps), **kwargs)


class FeedForward(nn.Module):
    """Linear -> GELU -> Dropout -> Linear -> Dropout (reference :20-31); `net` indices match the reference."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))
        self.dropout = dropout

    def _run(self, x, ln, residual):
        if self.dropout > 0 and self.training:
            return self._run_dropout(x, ln, residual)
        res_same = residual is x
        x = Fw._as_compute(x)
        if residual is not None:
            residual = x if res_same else Fw._as_compute(residual)
        l1, l2 = self.net[0], self.net[3]
        return Fw.feed_forward_block(x, ln, l1.weight, l1.bias, l2.weight, l2.bias, residual)

    def _run_dropout(self, x, ln, residual):
        """Training with dropout > 0 (local_3d_attention.py:20-31: Linear -> GELU -> Dropout -> Linear -> Dropout).  The reference's
        default and every published run use 0, so this is not a fused path: the two GEMMs are the library's (Fw.linear, weight
        gradients included), the LayerNorm, GELU, masks and the residual add are torch device ops between them."""
        F = torch.nn.functional
        x = Fw._as_compute(x)
        h = x if ln is None else F.layer_norm(x, (x.shape[-1],), ln[0].to(x.dtype), ln[1].to(x.dtype), 1e-5)
        l1, l2 = self.net[0], self.net[3]
        h = F.dropout(F.gelu(Fw.linear(h, l1.weight, l1.bias)), self.dropout, True)
        y = F.dropout(Fw.linear(h, l2.weight, l2.bias), self.dropout, True)
        return y if residual is None else y + Fw._as_compute(residual)

    def forward(self, x):
        return self._run(x, None, None).to(x.dtype)

    def forward_prenorm(self, x, norm, residual=None):
        return self._run(x, (norm.weight, norm.bias), residual)


class Local3dAttention(nn.Module):
    """Windowed 3D attention over a [B,S,H,W,dim] token grid (reference :34-118).

    `forward(x, q)`: keys/values are projected from x, queries from q; every token attends to its
    (2eS+1)(2eH+1)(2eW+1) neighbourhood clipped to the grid.  `use_checkpointing` is accepted for signature
    compatibility: the HIP forward keeps only the per-row log-sum-exp, so there is nothing to checkpoint."""

    def __init__(self, extents, dim, heads=8, dim_head=64, dropout=.0, use_checkpointing=True):
        super().__init__()
        self.extents = extents
        inner = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=True)
        if heads == 1 and dim_head == dim:
            self.to_out = nn.Identity()                                   # quirk Q6: no to_out.* keys
        else:
            self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        self.use_checkpointing = use_checkpointing
        self.dropout = dropout

    def _run(self, x, q, ln, residual, out_f32=False):
        same, res_same = q is x, residual is x            # attn(x, q=x) + x: keep ONE tensor so the backward folds the paths
        x = Fw._as_compute(x)
        q = x if same else Fw._as_compute(q)
        if residual is not None:
            residual = x if res_same else Fw._as_compute(residual)
        wq, wk, wv, bv = self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_v.bias
        if isinstance(self.to_out, nn.Identity):
            wo = bo = None
        else:
            wo, bo = self.to_out[0].weight, self.to_out[0].bias
        if (wq.shape[0] // self.heads) % 8 and wo is not None:
            wq, wk, wv, bv, wo = self._head_padded()
        if self.dropout > 0 and self.training and wo is not None:
            # to_out = Linear -> Dropout (local_3d_attention.py:50-53): the fused block without its residual, the mask and the
            # residual add as torch device ops behind it (not a fused path: the reference default and every published run use 0)
            y = Fw.attention_block(x, q, ln, wq, wk, wv, bv, wo, bo, None, self.extents, self.heads)
            y = torch.nn.functional.dropout(y, self.dropout, True)
            y = y if residual is None else y + residual
            return y.reshape(q.shape[:-1] + (y.shape[-1],))
        y = Fw.attention_block(x, q, ln, wq, wk, wv, bv, wo, bo, residual, self.extents, self.heads, out_f32=out_f32)
        return y.reshape(q.shape[:-1] + (y.shape[-1],))

    def _head_padded(self):
        """A dim_head that is no multiple of 8 (the kernels' 16-byte granule; the reference takes any --dim_head): every head of
        the projections zero-padded to the next multiple -- the padding contributes 0 to q . k and carries zeros through v and
        to_out's padded columns -- with the kernels' 1 / sqrt(padded width) corrected in to_q.  Built by torch ops on the
        parameters (a rare path): autograd carries the gradients back through the padding."""
        F = torch.nn.functional
        h = self.heads
        dh = self.to_q.weight.shape[0] // h
        pad = -dh % 8
        dim = self.to_q.weight.shape[1]

        def rows(w):
            return F.pad(w.view(h, dh, -1), (0, 0, 0, pad)).reshape(h * (dh + pad), -1)
        wq = rows(self.to_q.weight) * ((dh + pad) / dh) ** 0.5
        bv = F.pad(self.to_v.bias.view(h, dh), (0, pad)).reshape(-1)
        wo = F.pad(self.to_out[0].weight.view(dim, h, dh), (0, pad)).reshape(dim, h * (dh + pad))
        return wq, rows(self.to_k.weight), rows(self.to_v.weight), bv, wo

    def forward(self, x, q):
        return self._run(x, q, None, None, out_f32=q.dtype == torch.float32).to(q.dtype)

    def forward_prenorm(self, x, norm, q, residual=None):
        return self._run(x, q, (norm.weight, norm.bias), residual)

    def local_attention(self, k, v, q):
        """Attention core on already-projected tensors (reference :78-99), returned like the reference as
        [(b s h w), heads, 1, dim_head]."""
        from . import ops
        dt_in = q.dtype
        dh = q.shape[-1] // self.heads
        pad = -dh % 8
        if pad:                                            # (a head width off the 8-element granule: see _head_padded)
            F = torch.nn.functional

            def heads_padded(t, scale=1.0):
                return F.pad(t.reshape(t.shape[:-1] + (self.heads, dh)) * scale, (0, pad)).reshape(t.shape[:-1] + (-1,))
            q, k, v = heads_padded(q, ((dh + pad) / dh) ** 0.5), heads_padded(k), heads_padded(v)
        k, v, q = Fw._as_compute(k), Fw._as_compute(v), Fw._as_compute(q)
        out, _, _ = ops.local3d_attention_fwd(q, k, v, self.extents, self.heads)
        out = out.reshape(-1, self.heads, 1, out.shape[-1] // self.heads)
        return (out[..., :dh] if pad else out).to(dt_in)


class Local3dAttentionTransformer(nn.Module):
    """Token + 3-axis position embedding followed by depth x [attention, feed-forward] with residuals and no
    final LayerNorm (reference :121-163)."""

    def __init__(self, *, data_shape, dim, num_classes, extents, depth, heads, dim_head, mlp_dim, dropout=.0):
        super().__init__()
        self.num_classes = num_classes
        self.embedding = nn.Embedding(num_classes, dim)
        self.pos_emb_s = nn.Embedding(data_shape[0], dim)
        self.pos_emb_h = nn.Embedding(data_shape[1], dim)
        self.pos_emb_w = nn.Embedding(data_shape[2], dim)
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Local3dAttention(extents, dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)),
            ]))

    def get_pos_embedding(self, batch_shape):
        """(pos_s + pos_h) + pos_w broadcast over the grid (reference :140-151); fp32, for inspection."""
        _, s, h, w = batch_shape
        ps = self.pos_emb_s.weight[:s].view(1, s, 1, 1, -1)
        ph = self.pos_emb_h.weight[:h].view(1, 1, h, 1, -1)
        pw = self.pos_emb_w.weight[:w].view(1, 1, 1, w, -1)
        return ((ps + ph) + pw).expand(batch_shape[0], s, h, w, -1)

    def forward(self, img_z):
        """[B,S,H,W] int64 tokens -> [B,S,H,W,dim] in the parameters' dtype (see the module docstring's dtype rule)."""
        return self.forward_compute(img_z).to(self.embedding.weight.dtype)

    def check_grid(self, img_z):
        """The errors the reference raises for a grid / token id its embeddings cannot index."""
        _, S, H, W = img_z.shape
        if S > self.pos_emb_s.num_embeddings or H > self.pos_emb_h.num_embeddings or W > self.pos_emb_w.num_embeddings:
            raise IndexError('token grid larger than the position-embedding tables')
        from .config import get_check_tokens
        if get_check_tokens() and (int(img_z.min()) < 0 or int(img_z.max()) >= self.embedding.num_embeddings):
            raise IndexError('index out of range in self')            # what nn.Embedding raises in the reference

    def forward_compute(self, img_z):
        """forward() without the boundary cast: the residual stream in the compute dtype (internal callers)."""
        if not img_z.is_cuda:
            raise Fw.ops.L.WmzError('Local3dAttentionTransformer runs on the GPU only (no CPU fallback)')
        self.check_grid(img_z)
        if not torch.is_grad_enabled():
            from . import fused
            from .config import get_compute_dtype, get_fused_dtype
            if fused.supported(self, get_fused_dtype()):
                # inference, bf16 (or half: the precise mode), default widths: one attention launch + one per-token launch per
                # layer, the embedding fused into the first one
                return fused.transformer_forward(self, z=img_z)
            if fused.chain_supported(self, get_fused_dtype(), True) and (
                    fused.half_attention_ok(self, img_z.shape[2], img_z.shape[3]) if get_fused_dtype() == torch.float16
                    else fused.chain_pays(fused.chain_widths(self), img_z.numel(), False)):
                # the width table of csrc/chain_widths.h: the same fusion on csrc/layer_chain.hip -- its half unit in the precise
                # mode, where the planes are the row attention kernel's (other planes stay on the fp32 route below); in bfloat16
                # where the token count fills enough 128-token workgroups to beat the op-by-op GEMMs (fused.chain_pays)
                return fused.transformer_forward_chain(self, img_z)
        else:
            from . import config, fused
            if config.get_fused_training() and fused.supported(self, config.get_compute_dtype()):
                return fused.transformer_forward_train(self, img_z)     # training forward on the fused kernels
        x = Fw.embed_tokens(img_z, self.embedding.weight, self.pos_emb_s.weight, self.pos_emb_h.weight,
                            self.pos_emb_w.weight)
        for attn, ff in self.layers:
            # x = attn(x, q=x) + x ; x = ff(x) + x   with both residual adds fused into the GEMM epilogues
            x = attn.fn.forward_prenorm(x, attn.norm, q=x, residual=x)
            x = ff.fn.forward_prenorm(x, ff.norm, residual=x)
        return x
