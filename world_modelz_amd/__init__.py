"""world_modelz_amd -- MI355X-native (gfx950) denoiser hot path of world-modelz/vq-video-diffusion.

Drop-in nn.Module surfaces (same constructor signatures and state_dict keys as the reference):
    local_3d_attention.{PreNorm, FeedForward, Local3dAttention, Local3dAttentionTransformer}
    vq.VectorQuantizerEMA, autoencoder.{SimpleResidualEncoder, SimpleResidualDecoder},
    train_vqae.VqAutoEncoder, main.VqVideoDiffusionModel
whose forward/backward bodies call libwmz_hip.so (include/wmz.h) through ctypes.
GPU only: there is no CPU fallback (the CPU oracle under oracle/ is test infrastructure).
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
