"""Run-time configuration of the HIP path."""
import contextlib
import os

import torch

_DTYPES = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'fp32': torch.float32, 'float32': torch.float32,
           'f16': torch.float16, 'fp16': torch.float16, 'float16': torch.float16, 'precise': torch.float16}

# Activation / MFMA-operand dtype of the denoiser kernels.  Parameters stay fp32 (state_dict compatible with
# the reference); operand copies are cached per parameter version.  Three modes:
#   bf16    the speed mode named by BASELINE.json (end-to-end logits ~4e-3 from the fp32 reference on the default model);
#   fp32    the parity mode (exact-f32 MFMA at 1/16 of the matrix rate, op by op: ~4e-7, 11x slower);
#   float16 the PRECISE fused mode: inference at the default widths runs the fused kernels with IEEE-half MFMA operands and a half
#           residual stream between the layers (~5e-4 at the bf16 speed: inside the 1e-3 BASELINE.json asks for); everything the
#           half kernels are not built for (training, widths other than the default and the published ones, planes outside the row
#           attention kernel's shapes, the conv encoder / decoder) runs the fp32 route -- never bf16.
_precise = False
_compute_dtype = None


def get_compute_dtype():
    """The dtype of everything that is NOT on the fused inference kernels (fp32 in the precise mode)."""
    return _compute_dtype


def get_fused_dtype():
    """The operand / stream dtype of the fused inference kernels: torch.float16 in the precise mode, else the compute dtype."""
    return torch.float16 if _precise else _compute_dtype


def get_mode_dtype():
    """What set_compute_dtype was given (the context manager restores this)."""
    return torch.float16 if _precise else _compute_dtype


def set_compute_dtype(dt):
    global _compute_dtype, _precise
    if isinstance(dt, str):
        dt = _DTYPES[dt.lower()]
    if dt not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError(f'compute dtype must be float32, bfloat16 or float16 (the precise fused mode), got {dt}')
    _precise = dt == torch.float16
    _compute_dtype = torch.float32 if _precise else dt


set_compute_dtype(os.environ.get('WMZ_COMPUTE_DTYPE', 'bf16'))


@contextlib.contextmanager
def compute_dtype(dt):
    prev = get_mode_dtype()
    set_compute_dtype(dt)
    try:
        yield
    finally:
        set_compute_dtype(prev)


# Inference only: the denoiser returns the last frame's logits (reference main.py:33-36), so the planes outside the
# last frame's dependence cone are dead work.  With the cone on (default) they are not launched; the logits are
# bit-identical either way (tests/test_modules_gpu.py).  bench.py's headline figure runs with the cone OFF: it times
# the full grid, i.e. the same work the reference's forward does.
_last_frame_cone = os.environ.get('WMZ_LAST_FRAME_CONE', '1') != '0'


def get_last_frame_cone():
    return _last_frame_cone


def set_last_frame_cone(on):
    global _last_frame_cone
    _last_frame_cone = bool(on)


@contextlib.contextmanager
def last_frame_cone(on):
    prev = get_last_frame_cone()
    set_last_frame_cone(on)
    try:
        yield
    finally:
        set_last_frame_cone(prev)


# Training forward of the denoiser on the fused kernels (bf16, default widths): one attention launch + one per-token
# launch per layer instead of six; the backward recomputes the feed-forward pre-activation.  Off: the op-by-op forward.
_fused_training = os.environ.get('WMZ_FUSED_TRAINING', '1') != '0'


def get_fused_training():
    return _fused_training


def set_fused_training(on):
    global _fused_training
    _fused_training = bool(on)


# ... and its backward on the fused per-token backward kernels (csrc/layer_fused_bwd.hip: wmz_ff_fused_bwd /
# wmz_qkv_fused_bwd; the forward then also saves the feed-forward pre-activation).  Off: the op-by-op backward
# (backward.py) behind the fused forward.
# Which per-token path the chain kernels' widths (csrc/chain_widths.h) take in bfloat16: 'auto' = the chain kernels where they pay
# (fused.chain_pays: their workgroups hold 128 tokens and stream the layer's whole weight set each, so below a token count that
# grows with the weight bytes the op-by-op GEMMs -- tiled over tokens AND features -- are faster; profiles/r06/time_chain_tokens.txt),
# 'always' / 'never' for tests and timing tools.  The precise mode always takes the half chain kernels (its alternative is fp32).
_chain_policy = os.environ.get('WMZ_CHAIN_POLICY', 'auto')


def get_chain_policy():
    return _chain_policy


def set_chain_policy(policy):
    global _chain_policy
    if policy not in ('auto', 'always', 'never'):
        raise ValueError("chain policy: 'auto', 'always' or 'never'")
    prev, _chain_policy = _chain_policy, policy
    return prev


_fused_backward = True


def fused_backward():
    return _fused_backward


def set_fused_backward(on):
    global _fused_backward
    _fused_backward = bool(on)


# The reference's nn.Embedding raises IndexError on a token id outside the vocabulary; the HIP embedding clamps instead (a
# device-side check cannot raise without a host sync in the middle of the step).  With this switch on, the drop-in modules
# validate their token inputs on the host before launching (one device->host sync per call): the reference's error behaviour
# for debugging, off by default for throughput.
_check_tokens = os.environ.get('WMZ_CHECK_TOKENS', '0') != '0'


def get_check_tokens():
    return _check_tokens


def set_check_tokens(on):
    global _check_tokens
    _check_tokens = bool(on)


# Inference forward of the denoiser on the fused kernels: clips are independent, so the batch can be cut into groups whose
# launch chains run on parallel HIP streams (forked from / joined to the caller's stream; inside a hipGraph capture they become
# parallel branches of the graph).  Every kernel of the step is one workgroup per CU and a single wave of workgroups, so a
# full-batch launch pays its prologue burst, its critical path and its store tail with nothing to overlap them; two half-batch
# chains fill the chip with 128 workgroups each and overlap one chain's bursts and tails with the other's compute.  1 = off.
_clip_streams = int(os.environ.get('WMZ_CLIP_STREAMS', '2'))


# Training: weight gradients that land in the flat gradient arena are launched on a side stream (ops.linear_wgrad): they
# have no consumer inside the backward.  0 = on the compute stream.
_wgrad_stream = int(os.environ.get('WMZ_WGRAD_STREAM', '1'))


def get_wgrad_stream():
    return bool(_wgrad_stream)


def set_wgrad_stream(on):
    global _wgrad_stream
    _wgrad_stream = 1 if on else 0


def get_clip_streams():
    return _clip_streams


def set_clip_streams(n):
    global _clip_streams
    _clip_streams = max(1, int(n))


_shared_streams = {}


def shared_stream(kind, device=None):
    """ONE side stream per (purpose, device) for the life of the process.  torch hands out streams from a pool of 32 per device,
    round robin: a process that makes a new torch.cuda.Stream() for every capture warm-up / reducer / trainer (a test suite: ~250
    of them) sooner or later gets one that IS an older, still used stream -- the weight-gradient side stream, a clip chain, another
    trainer's reducer -- and two roles that the capture logic orders against each other become one queue (round 4: the full GPU
    suite died inside capture_end() of an RCCL-capturing step exactly when the number of capturing tests in front of it crossed
    such a wrap; any one group of them less and it passed)."""
    import torch
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = (kind, dev.index)
    s = _shared_streams.get(key)
    if s is None:
        s = _shared_streams[key] = torch.cuda.Stream(device=dev)
    return s

