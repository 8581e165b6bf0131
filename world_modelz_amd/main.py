"""MI355X drop-in for the model class of vq-video-diffusion/main.py (VqVideoDiffusionModel, :25-36)."""
import torch
from torch import nn

from . import functional as Fw
from .local_3d_attention import Local3dAttentionTransformer


class VqVideoDiffusionModel(nn.Module):
    """Denoiser: tokens [B,S,H,W] (vocabulary num_classes + 1, the extra id is the mask token) ->
    fp32 logits [B,H,W,num_classes] of the LAST frame only (reference :33-36; quirk Q5)."""

    def __init__(self, *, data_shape, dim, num_classes, extents, depth, dim_head, mlp_dim, heads=1, dropout=.0):
        super().__init__()
        self.transformer = Local3dAttentionTransformer(data_shape=data_shape, dim=dim, num_classes=num_classes + 1,
                                                       extents=extents, depth=depth, heads=heads, dim_head=dim_head,
                                                       mlp_dim=mlp_dim, dropout=dropout)
        self.logit_proj = nn.Linear(dim, num_classes)

    def forward(self, x):
        if not torch.is_grad_enabled() and x.is_cuda:
            from . import config, fused
            if config.get_last_frame_cone() and fused.supported(self.transformer, config.get_fused_dtype()):
                tr = self.transformer
                _, S, H, W = x.shape
                if S > tr.pos_emb_s.num_embeddings or H > tr.pos_emb_h.num_embeddings or W > tr.pos_emb_w.num_embeddings:
                    raise IndexError('token grid larger than the position-embedding tables')
                last = fused.transformer_forward_last(tr, x)      # only the planes the last frame depends on
                return self._logits(last)
        h = self.transformer.forward_compute(x)
        return self._logits(h[:, -1])         # [B,H,W,D] view: uniform row stride, no copy

    def _logits(self, last):
        # (a half stream -- the precise mode -- takes the half unit of the same GEMM: fp32 accumulation, fp32 logits)
        return Fw.linear(last, self.logit_proj.weight, self.logit_proj.bias, out_f32=True)
