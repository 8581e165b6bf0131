"""MI355X drop-in for the model side of minecraft/sparse_diffusion.py (config 5): VqSparseDiffusionModel :75-111 and
the position samplers :31-72.  The training script around them (MineRL loader, wandb) is out of scope."""
import math

import torch
import torch.nn as nn

from . import functional as Fw
from .transformer import Transformer


def sample_flat_positions(batch_size, context_length, s, h, w, device):
    """Uniform positions without replacement inside each pass over the grid (reference :31-41)."""
    max_index = s * h * w
    n = batch_size * context_length
    p = torch.empty(n, device=device, dtype=torch.long)
    j = 0
    while j < n:
        r = torch.randperm(max_index, device=device)
        take = min(n - j, max_index)
        p[j:j + take] = r[:take]
        j += take
    return p.view(batch_size, context_length)


def sample_time_dependent(batch_size, context_length, s, h, w, t, device, o=None):
    """Positions from a window of frames whose width grows with the diffusion time t (reference :44-72).  The
    reference loops over the batch with one randperm per item; here one batched top-k over random keys restricted to
    each item's window draws the same distribution (uniform without replacement) in a single device op."""
    t = t.reshape(-1).clamp(0, 1).to(device)
    assert context_length > 0
    min_sample_window = math.ceil(context_length / (h * w))
    assert min_sample_window < s
    sample_window = torch.floor(min_sample_window + (t * (s - min_sample_window + 1)))
    sample_window = sample_window.clamp(max=s - min_sample_window)
    if o is None:
        o = torch.rand_like(t)
    else:
        o = o.reshape(-1).clamp(0, 1 - 1e-5).to(device)
    offset = torch.floor(o * (s - sample_window + 1)).long() * h * w
    width = sample_window.long() * h * w                                   # [B] positions available per item
    keys = torch.rand(batch_size, s * h * w, device=device)
    keys = keys.masked_fill(torch.arange(s * h * w, device=device)[None, :] >= width[:, None], 2.0)
    idx = keys.topk(context_length, dim=1, largest=False).indices         # uniform without replacement in [0, width)
    return idx + offset[:, None]


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
        S, H, W = self.shape
        w_pos = indices % W
        h_pos = indices.div(W, rounding_mode='trunc') % H
        s_pos = indices.div(H * W, rounding_mode='trunc')
        return self.pos_emb_s(s_pos) + self.pos_emb_h(h_pos) + self.pos_emb_w(w_pos)

    def forward(self, x, indices):
        h = Fw.embed_tokens_indexed(x, indices, self.embedding.weight, self.pos_emb_s.weight, self.pos_emb_h.weight,
                                    self.pos_emb_w.weight, self.shape)
        h = self.transformer(h)
        return Fw.linear(h, self.logit_proj.weight, self.logit_proj.bias, out_f32=True)
