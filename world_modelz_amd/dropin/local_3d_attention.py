"""Shim: `from local_3d_attention import ...` in the reference scripts resolves to the MI355X classes."""
from world_modelz_amd.local_3d_attention import (FeedForward, Local3dAttention, Local3dAttentionTransformer,  # noqa: F401
                                                 PreNorm)
