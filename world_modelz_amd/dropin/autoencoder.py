"""Shim: `from autoencoder import ...` in the reference scripts resolves to the MI355X classes."""
from world_modelz_amd.autoencoder import (Residual, ResidualStack, SimpleResidualDecoder, SimpleResidualEncoder,  # noqa: F401
                                          UpscaleResidual, conv1x1, conv3x3)
