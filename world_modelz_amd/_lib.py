"""ctypes binding of libwmz_hip.so (the C ABI declared in include/wmz.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WMZ_LIB_PATH') or os.path.join(_HERE, 'libwmz_hip.so')    # override: kernel A/B builds (tools/)

WMZ_F32, WMZ_BF16, WMZ_F16 = 0, 1, 2
EXPECTED_VERSION = 113      # include/wmz.h WMZ_VERSION: bumped with every ABI change; lib() refuses another build
WMZ_LIN_GELU = 1
WMZ_LIN_GELU_IN = 2
WMZ_LIN_DGELU = 4

_lib = None

c_void_p, c_int, c_long, c_float, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float,
                                              ctypes.c_double)

# name -> argtypes, mirrors include/wmz.h one to one
SIGNATURES = {
    'wmz_local3d_attn_fwd': [c_void_p] * 6 + [c_int] * 9 + [c_long] * 4 + [c_int, c_void_p],
    'wmz_local3d_attn_fwd_general': [c_void_p] * 6 + [c_int] * 9 + [c_long] * 4 + [c_int, c_void_p],
    'wmz_debug_attn_knobs': [c_int, c_int],
    'wmz_debug_linear_knobs': [c_int],
    'wmz_local3d_attn_bwd': [c_void_p] * 10 + [c_int] * 9 + [c_long] * 8 + [c_int, c_void_p],
    'wmz_linear_wgrad': [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 4
                        + [c_int, c_int, c_void_p],
    'wmz_linear_wgrad_workspace_floats': [c_int, c_int, c_int, c_int],      # returns long (restype set in lib())
    'wmz_linear_wgrad_ws': [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 4
                           + [c_int, c_int, c_void_p, c_long, c_int, c_void_p],
    'wmz_linear_wgrad_batch': [c_int] + [c_void_p] * 12 + [c_long, c_int, c_void_p],
    'wmz_linear_wgrad_batch_ln': [c_int] + [c_void_p] * 14 + [c_void_p, c_long, c_int, c_void_p],
    'wmz_layernorm_stats': [c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p],
    'wmz_layernorm_bwd': [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p,
                          c_void_p, c_int, c_int, c_float, c_int, c_void_p],
    'wmz_embed_pos3d_bwd': [c_void_p] * 6 + [c_int] * 7 + [c_void_p],
    'wmz_embed_pos3d_bwd_workspace_ints': [c_int] * 5,                    # returns long
    'wmz_embed_pos3d_bwd_sorted': [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_long, c_int, c_void_p],
    'wmz_linear_fwd': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int,
                       c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p],
    'wmz_linear_fwd_stats': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p],
    'wmz_linear_fwd_gelu_pair': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p],
    'wmz_linear_fwd_train': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long,
                             c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p],
    'wmz_linear_fwd_blocked': [c_void_p, c_long, c_int, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int,
                               c_int, c_int, c_void_p],
    'wmz_embed_pos3d_fwd': [c_void_p] * 6 + [c_int] * 7 + [c_void_p],
    'wmz_debug_fused_timestamps': [c_void_p],
    'wmz_debug_attn_timestamps': [c_void_p],
    'wmz_debug_fused_knobs': [c_int],
    'wmz_debug_conv_knobs': [c_int, c_int],
    'wmz_debug_stamp': [c_void_p, c_int, c_void_p],
    'wmz_layer_fused_fwd': [c_void_p] * 7 + [c_int] * 6 + [c_float, c_void_p],
    'wmz_operands_refresh': [c_void_p] * 7 + [c_int, c_void_p],
    'wmz_conv_operands_refresh': [c_void_p] * 6 + [c_int, c_int, c_void_p],
    'wmz_conv_operands_refresh_packed': [c_void_p] * 7 + [c_int, c_int, c_void_p],
    'wmz_layer_fused_pack': [c_void_p] * 16 + [c_int] * 3 + [c_void_p],
    'wmz_layer_fused_fwd_train': [c_void_p] * 12 + [c_int] * 7 + [c_float, c_void_p],
    'wmz_fused_pack_table': [c_void_p, c_int, c_long, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_layer_fused_bwd_pack': [c_void_p] * 10 + [c_int] * 3 + [c_void_p],
    'wmz_ff_fused_bwd': [c_void_p] * 10 + [c_int] * 6 + [c_void_p, c_void_p],
    'wmz_qkv_fused_bwd': [c_void_p, c_long, c_void_p, c_long] + [c_void_p] * 6 + [c_int] * 3 + [c_void_p],
    'wmz_ln_affine_grads': [c_void_p] * 9 + [c_int] * 3 + [c_void_p],
    'wmz_ln_affine_grads_batch': [c_int] + [c_void_p] * 12 + [c_void_p],
    'wmz_embed_qkv_fused_fwd_train': [c_void_p] * 12 + [c_int] * 9 + [c_float, c_void_p],
    'wmz_layer_fused_fwd_planes': [c_void_p] * 7 + [c_int] * 10 + [c_float, c_void_p],
    'wmz_embed_qkv_fused_fwd_planes': [c_void_p] * 10 + [c_int] * 10 + [c_float, c_void_p],
    'wmz_layer_chain_supported': [c_int, c_int, c_int, c_void_p],
    'wmz_layer_chain_slab_pieces': [],
    'wmz_layer_chain_fwd_planes': [c_void_p] * 7 + [c_int] * 9 + [c_float, c_void_p],
    'wmz_layer_chain_fwd_train': [c_void_p] * 14 + [c_long] + [c_int] * 5 + [c_float, c_void_p],
    'wmz_chain_ff_bwd': [c_void_p] * 8 + [c_long] + [c_int] * 3 + [c_void_p],
    'wmz_chain_qkv_bwd': [c_void_p] * 7 + [c_long] + [c_int] * 2 + [c_void_p],
    'wmz_local3d_attn_fwd_planes': [c_void_p] * 5 + [c_int] * 9 + [c_long] * 4 + [c_int, c_int, c_int, c_void_p],
    'wmz_embed_qkv_fused_fwd': [c_void_p] * 10 + [c_int] * 8 + [c_float, c_void_p],
    'wmz_conv2d_nhwc_fwd': [c_void_p] * 9 + [c_int] * 10 + [c_float, c_int, c_void_p],
    'wmz_conv2d_nhwc_fwd_pre': [c_void_p] * 11 + [c_float] + [c_int] * 10 + [c_float, c_int, c_void_p],
    'wmz_conv3x3_direct_supported': [c_int] * 4,
    'wmz_conv3x3_direct_pack_elems': [c_int, c_int],                       # returns long
    'wmz_conv3x3_direct_pack': [c_void_p, c_void_p, c_int, c_int, c_void_p],
    'wmz_conv3x3_direct_fwd': [c_void_p] * 9 + [c_int] * 6 + [c_float, c_void_p],
    'wmz_conv3x3_direct_supported_strided': [c_int] * 5,
    'wmz_conv3x3_direct_fwd_strided': [c_void_p] * 9 + [c_int] * 7 + [c_float, c_void_p],
    'wmz_conv_point_supported': [c_int] * 9,
    'wmz_conv_point_pack_elems': [c_int, c_int],                           # returns long
    'wmz_conv_point_pack': [c_void_p, c_void_p, c_int, c_int, c_void_p],
    'wmz_conv_point_fwd': [c_void_p] * 10 + [c_float] + [c_int] * 10 + [c_float, c_void_p],
    'wmz_conv_point_fwd_bn': [c_void_p] * 11 + [c_float] + [c_int] * 10 + [c_float, c_void_p],      # (.., in_shift, const wmz_bn_stats*, in_slope, ..)
    'wmz_affine_act_bn_supported': [c_int, c_int],
    'wmz_dilate_nhwc': [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p],
    'wmz_affine_act_nhwc_bn': [c_void_p] * 9 + [c_long, c_int, c_int, c_float, c_int, c_void_p],
    'wmz_nchw_to_nhwc8': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_conv2d_nhwc_wgrad_batch': [c_int] + [c_void_p] * 17 + [c_long, c_int, c_void_p],
    'wmz_conv2d_nhwc_wgrad_is_direct': [c_int] * 10,
    'wmz_channel_stats_nhwc': [c_void_p, c_long, c_int, c_void_p, c_void_p, c_int, c_void_p],
    'wmz_bn_finalize': [c_void_p, c_void_p, c_double] + [c_void_p] * 4 + [c_double, c_double, c_int] + [c_void_p] * 4
                       + [c_int, c_void_p, c_void_p],
    'wmz_conv2d_nhwc_wgrad': [c_void_p] * 4 + [c_int] * 10 + [c_void_p],
    'wmz_conv2d_nhwc_wgrad_workspace_floats': [c_int] * 10,                # returns long
    'wmz_conv2d_nhwc_wgrad_ws': [c_void_p] * 4 + [c_int] * 12 + [c_void_p, c_long, c_int, c_void_p],
    'wmz_bn_act_bwd_reduce': [c_void_p] * 8 + [c_long, c_int, c_int, c_float, c_int, c_void_p],
    'wmz_bn_bwd_apply': [c_void_p] * 8 + [c_long, c_int, c_int, c_void_p],
    'wmz_bn_leaky_bwd_supported': [c_int, c_int],
    'wmz_bn_leaky_bwd': [c_void_p] * 11 + [c_long, c_int, c_float, c_int, c_void_p],
    'wmz_bn_bwd_apply_add': [c_void_p] * 9 + [c_long, c_int, c_int, c_void_p],
    'wmz_bilinear2x_nhwc_bwd': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_affine_act_nhwc': [c_void_p] * 7 + [c_long, c_int, c_int, c_float, c_int, c_void_p],
    'wmz_bilinear2x_nhwc': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_embed_indexed_fwd': [c_void_p] * 7 + [c_long] + [c_int] * 6 + [c_void_p],
    'wmz_embed_indexed_bwd': [c_void_p] * 7 + [c_long] + [c_int] * 6 + [c_void_p],
    'wmz_corrupt_tokens': [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_int,
                           ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p],
    'wmz_loss_partials_workspace_floats': [],
    'wmz_vq_tail_fwd': [c_void_p] * 7 + [c_long, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_vq_tail_bwd': [c_void_p] * 5 + [c_long, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_recon_loss_fwd': [c_void_p] * 4 + [c_long, c_long, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_recon_loss_bwd': [c_void_p] * 4 + [c_long, c_long, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_sparse_draw_context_supported': [c_int, c_int, c_int],
    'wmz_sparse_draw_context': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, c_void_p],
    'wmz_categorical_scatter': [c_void_p, c_long, c_long, c_int, c_void_p, c_void_p, c_long, c_long, c_void_p, ctypes.c_ulonglong,
                                ctypes.c_ulonglong, c_void_p, c_void_p],
    'wmz_corrupt_tokens_dev': [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_int,
                               ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, c_void_p],
    'wmz_sample_tokens_dev': [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_long, c_long, c_void_p,
                              c_void_p, ctypes.c_ulonglong, c_void_p, c_void_p],
    'wmz_adamw_step_dev': [c_void_p] * 4 + [c_long, c_void_p] + [c_double] * 5 + [c_void_p, c_void_p],
    'wmz_ce_fwd': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    'wmz_ce_bwd': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    'wmz_ce_fwd_bwd': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    'wmz_grad_sqnorm': [c_void_p, c_long, c_float, c_void_p, c_void_p],
    'wmz_adamw_step': [c_void_p] * 4 + [c_long] + [c_double] * 5 + [c_long, c_double, c_void_p],
    'wmz_vq_argmin': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'wmz_vq_argmin_screened_workspace_bytes': [c_int, c_int, c_int],        # returns long
    'wmz_vq_argmin_screened': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p],
    'wmz_vq_gather': [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p],
    'wmz_vq_ema_stats': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                         c_void_p],
    'wmz_vq_ema_stats_workspace_ints': [c_int, c_int],                    # returns long
    'wmz_vq_ema_stats_sorted': [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_long, c_void_p],
    'wmz_vq_ema_update': [c_void_p] * 5 + [c_int, c_int, c_double, c_double, c_void_p],
}


for _n in ('wmz_layer_fused_fwd', 'wmz_embed_qkv_fused_fwd', 'wmz_layer_fused_fwd_planes', 'wmz_embed_qkv_fused_fwd_planes',
           'wmz_layer_fused_pack', 'wmz_fused_pack_table', 'wmz_linear_fwd', 'wmz_linear_fwd_stats', 'wmz_linear_fwd_blocked',
           'wmz_layer_chain_fwd_planes'):
    SIGNATURES[_n + '_f16'] = SIGNATURES[_n]          # the precise (IEEE half) instantiations: include/wmz.h


class WmzError(RuntimeError):
    pass


class BnStats(ctypes.Structure):
    """include/wmz.h wmz_bn_stats: a host struct of device pointers, read by the call."""
    _fields_ = [('sum', c_void_p), ('sq', c_void_p), ('gamma', c_void_p), ('beta', c_void_p), ('running_mean', c_void_p),
                ('running_var', c_void_p), ('num_batches_tracked', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
                ('mean', c_void_p), ('rstd', c_void_p), ('count', c_double), ('momentum', c_double), ('eps', c_double)]


def lib():
    """Load (once) and return the shared library; raises WmzError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WmzError(f'{LIB_PATH} not found: build it with `python -m world_modelz_amd.build` '
                           '(there is no CPU fallback for the HIP path)')
        L = ctypes.CDLL(LIB_PATH)
        L.wmz_version.restype = c_int
        if L.wmz_version() != EXPECTED_VERSION:
            raise WmzError(f'{LIB_PATH} is version {L.wmz_version()}, this Python package expects {EXPECTED_VERSION} (include/wmz.h '
                           'WMZ_VERSION): a stale build -- run `python -m world_modelz_amd.build`')
        L.wmz_last_error.restype = ctypes.c_char_p
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:
                continue  # declared but not built yet: calling it raises below
            fn.argtypes = argtypes
            fn.restype = c_long if name.endswith(('_workspace_floats', '_workspace_ints', '_workspace_bytes', '_pack_elems')) else c_int
        _lib = L
    return _lib


after_call = None      # ops: launches deferred until the compute stream's NEXT launch has been issued (ops._flush_deferred)


def call(name, *args):
    L = lib()
    fn = getattr(L, name, None)
    if fn is None:
        raise WmzError(f'{name} is not exported by {LIB_PATH}')
    rc = fn(*args)
    if rc != 0:
        raise WmzError(f'{name} failed (code {rc}): {L.wmz_last_error().decode()}')
    if after_call is not None:
        after_call()


def dtype_code(dt):
    if dt == torch.float32:
        return WMZ_F32
    if dt == torch.bfloat16:
        return WMZ_BF16
    if dt == torch.float16:
        return WMZ_F16          # (the precise fused inference mode: the few entry points that take it say so in include/wmz.h)
    raise WmzError(f'unsupported activation dtype {dt}: the HIP path computes in float32 or bfloat16')


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  CPU tensors are refused: the product path is GPU-only."""
    if t is None:
        return None
    if not t.is_cuda:
        raise WmzError('libwmz_hip.so needs device (ROCm) tensors; got a CPU tensor '
                       '(the CPU oracle lives under oracle/ and is test infrastructure only)')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
