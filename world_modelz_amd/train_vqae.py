"""MI355X drop-in for the model class of vq-video-diffusion/train_vqae.py (VqAutoEncoder, :22-55)."""
import torch
import torch.nn as nn

from .autoencoder import SimpleResidualDecoder, SimpleResidualEncoder, _pad8
from .config import get_compute_dtype
from .vq import VectorQuantizerEMA


class VqAutoEncoder(nn.Module):
    """encoder -> VectorQuantizerEMA -> decoder.  The encoder emits NHWC latents straight into the codebook search
    and the gathered rows go straight into the decoder: the reference's BCHW<->BHWC permutes (:37-41, :47, :53) are
    layout no-ops here."""

    def __init__(self, embedding_dim, num_embeddings, downscale_steps=2, hidden_planes=128, in_channels=3):
        super().__init__()
        self.encoder = SimpleResidualEncoder(in_channels, embedding_dim, downscale_steps, hidden_planes)
        decoder_cfg = [hidden_planes for _ in range(downscale_steps)]
        self.decoder = SimpleResidualDecoder(decoder_cfg, in_channels=embedding_dim, out_channels=in_channels)
        self.vq = VectorQuantizerEMA(embedding_dim, num_embeddings)

    def _latents(self, x):
        return self.encoder.forward_nhwc(x).float()          # [B,h,w,E]; the codebook search is fp32

    def _decode_latents(self, q):
        E = q.shape[-1]
        if _pad8(E) != E:
            q = torch.nn.functional.pad(q, (0, _pad8(E) - E))
        return self.decoder.forward_nhwc(q.to(get_compute_dtype()).contiguous()).float()

    def _quantized_nhwc(self, x):
        """frames -> (straight-through latents in the decoder's operand dtype, channel-padded for its first conv; commitment loss;
        perplexity): the encoder's NHWC output goes to the quantiser as it is, whose tail kernel writes the decoder's operand."""
        h = self.encoder.forward_nhwc(x)
        st, _, latent_loss, perplexity = self.vq.forward_fused(h, get_compute_dtype(), True)
        return st, latent_loss, perplexity

    def forward(self, x):
        st, latent_loss, perplexity = self._quantized_nhwc(x)
        return self.decoder.forward_nhwc(st).float(), latent_loss, perplexity

    def training_losses(self, x, loss_kind):
        """(reconstruction loss, commitment loss, perplexity) of train_vqae.py:139-150 without the reconstruction as an NCHW fp32
        tensor: the loss (ops.recon_loss: 'SmoothL1' | 'MSE' | 'MAE' | 'L1', reduction mean) reads the decoder's NHWC output in
        place and its backward writes that layout.  x: NCHW fp32 frames on the GPU."""
        from . import ops
        st, latent_loss, perplexity = self._quantized_nhwc(x)
        y = self.decoder.forward_nhwc(st, raw=True)
        return ops.recon_loss(y, x.contiguous(), loss_kind), latent_loss, perplexity

    def encode(self, x):
        h = self._latents(x)
        return self.vq.encode(h).view(h.shape[:-1])

    def decode(self, z):
        return self._decode_latents(self.vq.decode(z))
