"""Run-time configuration of the HIP path."""
import contextlib
import os

import torch

_DTYPES = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'fp32': torch.float32, 'float32': torch.float32}

# Activation / MFMA-operand dtype of the denoiser kernels.  Parameters stay fp32 (state_dict compatible with
# the reference); bf16 operand copies are cached per parameter version.  fp32 is the parity mode
# (exact-f32 MFMA), bf16 the speed mode named by BASELINE.json.
_compute_dtype = _DTYPES[os.environ.get('WMZ_COMPUTE_DTYPE', 'bf16').lower()]


def get_compute_dtype():
    return _compute_dtype


def set_compute_dtype(dt):
    global _compute_dtype
    if isinstance(dt, str):
        dt = _DTYPES[dt.lower()]
    if dt not in (torch.float32, torch.bfloat16):
        raise ValueError(f'compute dtype must be float32 or bfloat16, got {dt}')
    _compute_dtype = dt


@contextlib.contextmanager
def compute_dtype(dt):
    prev = get_compute_dtype()
    set_compute_dtype(dt)
    try:
        yield
    finally:
        set_compute_dtype(prev)
