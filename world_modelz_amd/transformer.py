"""MI355X drop-in for minecraft/transformer.py (lucidrains ViT blocks used by config 5): PreNorm :11-17,
FeedForward :19-31, Attention :33-63, Transformer :66-80.  Same constructors and state_dict keys.

Dense softmax attention over n tokens is local 3D attention whose window covers the whole grid, so it runs on the
same HIP attention kernels (n % 16 == 0 takes the 16-wide-plane fast path); the fused to_qkv GEMM carries the PreNorm
LayerNorm in its prologue and to_out carries the residual add in its epilogue.
"""
import torch
from torch import nn

from . import functional as Fw
from .local_3d_attention import FeedForward, PreNorm  # same modules as the local-attention stack  # noqa: F401


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        if heads == 1 and dim_head == dim:
            self.to_out = nn.Identity()
        else:
            self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
        self.p_drop = dropout

    def _run(self, x, ln, residual):
        if self.p_drop > 0 and self.training:
            return self._run_dropout(x, ln, residual)
        res_same = residual is x
        x = Fw._as_compute(x)
        if residual is not None:
            residual = x if res_same else Fw._as_compute(residual)
        wo, bo = (None, None) if isinstance(self.to_out, nn.Identity) else (self.to_out[0].weight, self.to_out[0].bias)
        wqkv = self.to_qkv.weight
        if (wqkv.shape[0] // (3 * self.heads)) % 8 and wo is not None:
            wqkv, wo = self._head_padded()
        return Fw.dense_attention_block(x, ln, wqkv, wo, bo, residual, self.heads)

    def _head_padded(self):
        """A dim_head off the kernels' 8-element granule: every head of q | k | v zero-padded to the next multiple, the softmax
        scale of the padded width corrected in the q rows, to_out's columns padded to match (local_3d_attention.py::_head_padded:
        torch ops on the parameters, gradients through autograd)."""
        F = torch.nn.functional
        h = self.heads
        w = self.to_qkv.weight
        dh = w.shape[0] // (3 * h)
        pad = -dh % 8
        dim = w.shape[1]
        w3 = F.pad(w.view(3, h, dh, dim), (0, 0, 0, pad))                       # [3, h, dh + pad, dim]
        scale = torch.ones(3, 1, 1, 1, device=w.device, dtype=w.dtype)
        scale[0] = ((dh + pad) / dh) ** 0.5
        wo = F.pad(self.to_out[0].weight.view(dim, h, dh), (0, pad)).reshape(dim, h * (dh + pad))
        return (w3 * scale).reshape(3 * h * (dh + pad), dim), wo

    def _run_dropout(self, x, ln, residual):
        """Training with dropout > 0 (transformer.py:44-62: softmax -> Dropout on the attention PROBABILITIES -> . V -> to_out ->
        Dropout).  The reference default (sparse_diffusion.py:76) is 0, so this is not a fused path: the probabilities have to exist
        to be masked, which the flash-style HIP attention never lets them -- the two projections are the library's GEMMs
        (Fw.linear, weight gradients included); LayerNorm, the n x n scores, softmax, both masks, the value product and the
        residual add are torch device ops (n = 512 context tokens at config 5)."""
        F = torch.nn.functional
        x = Fw._as_compute(x)
        h = x if ln is None else F.layer_norm(x, (x.shape[-1],), ln[0].to(x.dtype), ln[1].to(x.dtype), 1e-5)
        B, n = h.shape[0], h.shape[1]
        qkv = Fw.linear(h, self.to_qkv.weight, None).view(B, n, 3, self.heads, -1).permute(2, 0, 3, 1, 4)     # [3, B, heads, n, dh]
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = torch.softmax(torch.matmul(q, k.transpose(-1, -2)).float() * self.scale, dim=-1)
        attn = F.dropout(attn, self.p_drop, True).to(v.dtype)
        o = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(B, n, -1)
        if not isinstance(self.to_out, nn.Identity):
            o = F.dropout(Fw.linear(o, self.to_out[0].weight, self.to_out[0].bias), self.p_drop, True)
        return o if residual is None else o + Fw._as_compute(residual)

    def forward(self, x):
        return self._run(x, None, None).to(x.dtype)

    def forward_prenorm(self, x, norm, residual=None):
        return self._run(x, (norm.weight, norm.bias), residual)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)),
            ]))

    def forward(self, x):
        return self.forward_compute(x).to(x.dtype)       # boundary dtype rule: see local_3d_attention.py

    def forward_compute(self, x):
        x = Fw._as_compute(x)
        for attn, ff in self.layers:
            x = attn.fn.forward_prenorm(x, attn.norm, residual=x)     # attn(x) + x
            x = ff.fn.forward_prenorm(x, ff.norm, residual=x)         # ff(x) + x
        return x
