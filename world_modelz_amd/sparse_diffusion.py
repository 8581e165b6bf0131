"""MI355X drop-in for the model side of minecraft/sparse_diffusion.py (config 5): VqSparseDiffusionModel :75-111 and
the position samplers :31-72.  The training script around them (MineRL loader, wandb) is out of scope."""
import math

import torch
import torch.nn as nn

from . import functional as Fw
from .transformer import Transformer


def _ranks_of_random_keys(rows, cols, device, generator=None):
    """[rows, cols] int64: row i is a uniform random permutation of 0..cols-1 (argsort of i.i.d. uniform keys)."""
    return torch.rand(rows, cols, device=device, generator=generator).argsort(dim=1)


def sample_flat_positions(batch_size, context_length, s, h, w, device):
    """batch_size * context_length flat grid positions, laid out as consecutive passes over the s*h*w grid, every pass a
    uniform permutation without replacement (semantics of reference :31-41).  One batched argsort instead of the
    reference's randperm-per-pass loop: pass k of the flattened request is row k of the permutation matrix."""
    grid, want = s * h * w, batch_size * context_length
    passes = -(-want // grid)
    perms = _ranks_of_random_keys(passes, grid, device)
    return perms.reshape(-1)[:want].reshape(batch_size, context_length)


def time_window(context_length, s, h, w, t, o=None):
    """Frame window (first frame, number of frames) per batch item for diffusion time t in [0, 1] (reference :53-64):
    the window spans at least ceil(context_length / (h w)) frames, grows linearly with t and is placed by o in [0, 1)
    (uniform when None).  Returns int64 tensors (first, frames)."""
    t = t.reshape(-1).clamp(0, 1)
    need = -(-context_length // (h * w))                     # frames that hold context_length positions
    assert context_length > 0 and need < s
    frames = (need + t * (s - need + 1)).floor().clamp(max=s - need)
    o = torch.rand_like(t) if o is None else o.reshape(-1).to(t.device).clamp(0, 1 - 1e-5)
    first = (o * (s - frames + 1)).floor()
    return first.long(), frames.long()


def sample_time_dependent(batch_size, context_length, s, h, w, t, device, o=None):
    """context_length distinct positions per item, uniform inside the item's frame window (reference :44-72).  The
    reference draws one randperm per batch item in a Python loop; here every item ranks i.i.d. uniform keys over the
    grid, keys outside its window pushed past 1, and keeps the context_length smallest -- the same law (uniform without
    replacement inside the window), one device op for the whole batch."""
    first, frames = time_window(context_length, s, h, w, t.to(device), o)
    grid = s * h * w
    inside = torch.arange(grid, device=device).unsqueeze(0) < (frames * (h * w)).unsqueeze(1)
    keys = torch.where(inside, torch.rand(batch_size, grid, device=device), torch.full((), 2.0, device=device))
    picked = keys.topk(context_length, dim=1, largest=False).indices
    return picked + (first * (h * w)).unsqueeze(1)


class VqSparseDiffusionModel(nn.Module):
    """tokens [B,n] (vocabulary num_classes + 1) at flat grid positions `indices` [B,n] -> logits [B,n,num_classes]."""

    def __init__(self, *, shape, dim, num_classes, depth, dim_head, mlp_dim, heads=1, dropout=0.0):
        super().__init__()
        self.shape = shape
        S, H, W = shape
        self.pos_emb_s = nn.Embedding(S, dim)
        self.pos_emb_h = nn.Embedding(H, dim)
        self.pos_emb_w = nn.Embedding(W, dim)
        self.embedding = nn.Embedding(num_classes + 1, dim)
        self.transformer = Transformer(dim=dim, depth=depth, heads=heads, dim_head=dim_head, mlp_dim=mlp_dim,
                                       dropout=dropout)
        self.logit_proj = nn.Linear(dim, num_classes)

    def pos_embedding_3d(self, indices):
        """fp32 position embedding of flat indices (reference :101-105); inspection helper, torch ops."""
        _, H, W = self.shape
        plane, in_plane = torch.div(indices, H * W, rounding_mode='floor'), indices.remainder(H * W)
        row, col = torch.div(in_plane, W, rounding_mode='floor'), in_plane.remainder(W)
        tables = (self.pos_emb_s.weight, self.pos_emb_h.weight, self.pos_emb_w.weight)
        return sum(tab[ix] for tab, ix in zip(tables, (plane, row, col)))

    def forward(self, x, indices):
        h = Fw.embed_tokens_indexed(x, indices, self.embedding.weight, self.pos_emb_s.weight, self.pos_emb_h.weight,
                                    self.pos_emb_w.weight, self.shape)
        h = self.transformer.forward_compute(h)
        return Fw.linear(h, self.logit_proj.weight, self.logit_proj.bias, out_f32=True)
