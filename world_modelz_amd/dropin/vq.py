"""Shim: `from vq import VectorQuantizerEMA` in the reference scripts resolves to the MI355X class."""
from world_modelz_amd.vq import VectorQuantizerEMA  # noqa: F401
