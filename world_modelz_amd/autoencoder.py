"""MI355X drop-in for vq-video-diffusion/autoencoder.py (conv3x3/conv1x1 :8-15, Residual :18-42, ResidualStack
:45-57, SimpleResidualEncoder :60-86, UpscaleResidual :89-131, SimpleResidualDecoder :134-152).

Same classes, constructor signatures, attribute names and state_dict keys (the parameter-holding sub-modules are
ordinary nn.Conv2d / nn.BatchNorm2d, created in the reference's order so that seeded initialisation matches).  The
forward passes run NHWC implicit-GEMM convolutions on MFMA with BatchNorm / LeakyReLU / skip adds fused into the
epilogues (eval mode) or applied by one elementwise kernel after the batch statistics are known (training mode,
which is what the frozen AE runs in inside main.py -- quirk Q3: running statistics are updated even under no_grad).
Tensors cross the module boundary as logical NCHW in channels_last memory format: no layout copies inside.
Forward only this round: the conv backward (VQ-AE training) is not built yet and raises.
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _cast, ops
from ._lib import WmzError
from .config import get_compute_dtype

LEAKY = 0.01


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups, bias=bias,
                     dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=bias)


def _pad8(c):
    return (c + 7) // 8 * 8


def _w_op(conv, dtype):
    """nn.Conv2d weight [Cout,Cin,KH,KW] -> GEMM operand [Cout, KH*KW*Cin8] (tap-major, channel fastest, zero-padded)."""
    def build(w):
        co, ci, kh, kw = w.shape
        w = w.permute(0, 2, 3, 1)
        if _pad8(ci) != ci:
            w = F.pad(w, (0, _pad8(ci) - ci))
        return w.reshape(co, -1)
    return _cast.operand((conv.weight,), dtype, 'conv', build)


def _to_nhwc(x, dtype):
    """logical NCHW (any memory format) -> contiguous [B,H,W,C8] in the compute dtype."""
    if not x.is_cuda:
        raise WmzError('the conv encoder/decoder runs on the GPU only (no CPU fallback)')
    x = x.permute(0, 2, 3, 1)
    c = x.shape[-1]
    if _pad8(c) != c:
        x = F.pad(x, (0, _pad8(c) - c))
    return x.to(dtype).contiguous()


def _to_nchw_view(y):
    """[B,H,W,C] contiguous -> logical NCHW view (channels_last memory format, no copy)."""
    return y.permute(0, 3, 1, 2)


def _check_no_grad(*mods):
    if torch.is_grad_enabled() and any(p.requires_grad for m in mods for p in m.parameters()):
        raise NotImplementedError('conv encoder/decoder backward is not built yet on the HIP path: run the '
                                  'auto-encoder under torch.no_grad() (the denoiser trains against a frozen AE)')


def _conv(x, conv, dtype, **kw):
    k = conv.kernel_size[0]
    bias = conv.bias.detach() if conv.bias is not None else None
    return ops.conv2d_nhwc(x, _w_op(conv, dtype), k, k, conv.stride[0], conv.padding[0], bias=bias, **kw)


def _count(t):
    return t.numel() // t.shape[-1]


class Residual(nn.Module):
    def __init__(self, in_planes, hidden_planes, stride=1, normalize=nn.BatchNorm2d, nonlinearity=nn.LeakyReLU):
        super().__init__()
        self._block = nn.Sequential(conv3x3(in_planes, hidden_planes, stride=stride), normalize(hidden_planes),
                                    nonlinearity(inplace=True), conv1x1(hidden_planes, in_planes), normalize(in_planes))
        if stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, in_planes, kernel_size=stride, stride=stride, bias=False),
                                            normalize(in_planes))
        else:
            self.downsample = None

    def forward_nhwc(self, x, dtype):
        c1, bn1, c2, bn2 = self._block[0], self._block[1], self._block[3], self._block[4]
        if bn1.training:
            h, s, q = _conv(x, c1, dtype, stats=True)
            sc, sh = ops.bn_finalize(bn1, s, q, _count(h))
            h = ops.affine_act_nhwc(h, sc, sh, leaky=True, slope=LEAKY)
            h, s, q = _conv(h, c2, dtype, stats=True)
            sc2, sh2 = ops.bn_finalize(bn2, s, q, _count(h))
            if self.downsample is not None:
                r, s, q = _conv(x, self.downsample[0], dtype, stats=True)
                sc3, sh3 = ops.bn_finalize(self.downsample[1], s, q, _count(r))
                return ops.affine_act_nhwc(h, sc2, sh2, r, sc3, sh3, leaky=True, slope=LEAKY)
            return ops.affine_act_nhwc(h, sc2, sh2, x, leaky=True, slope=LEAKY)
        # eval: BatchNorm folds into the conv epilogues, the skip add + LeakyReLU into the 1x1 conv's
        sc, sh = ops.bn_finalize(bn1, None, None, 0)
        h = _conv(x, c1, dtype, scale=sc, shift=sh, leaky=True, slope=LEAKY)
        if self.downsample is not None:
            sc3, sh3 = ops.bn_finalize(self.downsample[1], None, None, 0)
            r = _conv(x, self.downsample[0], dtype, scale=sc3, shift=sh3)
        else:
            r = x
        sc2, sh2 = ops.bn_finalize(bn2, None, None, 0)
        return _conv(h, c2, dtype, scale=sc2, shift=sh2, residual=r, leaky=True, slope=LEAKY)

    def forward(self, x):
        _check_no_grad(self)
        dt = get_compute_dtype()
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x, dt), dt))


class ResidualStack(nn.Module):
    def __init__(self, in_planes, num_layers, hidden_planes):
        super().__init__()
        self._num_residual_layers = num_layers
        layers = []
        for _ in range(num_layers):
            layers.append(Residual(in_planes, hidden_planes, 1))
            layers.append(Residual(in_planes, hidden_planes, 2))
        self._stack = nn.Sequential(*layers)

    def forward_nhwc(self, x, dtype):
        for blk in self._stack:
            x = blk.forward_nhwc(x, dtype)
        return x

    def forward(self, x):
        _check_no_grad(self)
        dt = get_compute_dtype()
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x, dt), dt))


class SimpleResidualEncoder(nn.Module):
    def __init__(self, in_planes, out_planes, num_layers, hidden_planes):
        super().__init__()
        self._conv_1 = conv3x3(in_planes, out_planes)
        self._residual_stack = ResidualStack(in_planes=out_planes, num_layers=num_layers, hidden_planes=hidden_planes)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_nhwc(self, x):
        """NCHW frames -> [B,h,w,E] latents (what VectorQuantizerEMA wants: no NCHW<->NHWC flips)."""
        _check_no_grad(self)
        dt = get_compute_dtype()
        h = _conv(_to_nhwc(x, dt), self._conv_1, dt, leaky=True, slope=LEAKY)
        return self._residual_stack.forward_nhwc(h, dt)

    def forward(self, x):
        return _to_nchw_view(self.forward_nhwc(x))


class UpscaleResidual(nn.Module):
    def __init__(self, in_planes, out_planes, upsample):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, out_planes, kernel_size=3, padding=1, bias=True)
        self.conv2 = nn.Conv2d(out_planes, out_planes, kernel_size=3, padding=1, bias=True)
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.bn2 = nn.BatchNorm2d(out_planes)
        self.act1 = nn.LeakyReLU(inplace=True)
        self.act2 = nn.LeakyReLU(inplace=True)
        self.upsample = upsample
        self.learn_conv_residual = in_planes != out_planes or upsample
        if self.learn_conv_residual:
            self.conv_residual = nn.Conv2d(in_planes, out_planes, kernel_size=1, padding=0)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_nhwc(self, x, dtype):
        if self.bn1.training:
            s, q = ops.channel_stats_nhwc(x)
            sc, sh = ops.bn_finalize(self.bn1, s, q, _count(x))
        else:
            sc, sh = ops.bn_finalize(self.bn1, None, None, 0)
        h = ops.affine_act_nhwc(x, sc, sh, leaky=True, slope=LEAKY)
        if self.upsample:
            h = ops.bilinear2x_nhwc(h)
            x = ops.bilinear2x_nhwc(x)
        if self.bn2.training:
            h, s, q = _conv(h, self.conv1, dtype, stats=True)
            sc, sh = ops.bn_finalize(self.bn2, s, q, _count(h))
            h = ops.affine_act_nhwc(h, sc, sh, leaky=True, slope=LEAKY)
        else:
            sc, sh = ops.bn_finalize(self.bn2, None, None, 0)
            h = _conv(h, self.conv1, dtype, scale=sc, shift=sh, leaky=True, slope=LEAKY)
        if self.learn_conv_residual:
            x = _conv(x, self.conv_residual, dtype)
        return _conv(h, self.conv2, dtype, residual=x)

    def forward(self, x):
        _check_no_grad(self)
        dt = get_compute_dtype()
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x, dt), dt))


class SimpleResidualDecoder(nn.Module):
    def __init__(self, cfg, in_channels, out_channels=3):
        super().__init__()
        upsample = functools.partial(F.interpolate, scale_factor=2, mode='bilinear', align_corners=False)
        layers = [conv3x3(in_channels, in_channels)]
        for hidden_channels in cfg:
            layers += [UpscaleResidual(in_channels, hidden_channels, upsample)]
            in_channels = hidden_channels
        layers += [conv3x3(in_channels, out_channels)]
        self.decoder_stack = nn.Sequential(*layers)

    def forward_nhwc(self, h):
        """[B,h,w,E] latents (NHWC, C % 8 == 0) -> logical NCHW image."""
        _check_no_grad(self)
        dt = get_compute_dtype()
        mods = list(self.decoder_stack)
        h = _conv(h, mods[0], dt)
        for m in mods[1:-1]:
            h = m.forward_nhwc(h, dt)
        return _to_nchw_view(_conv(h, mods[-1], dt))

    def forward(self, x):
        return self.forward_nhwc(_to_nhwc(x, get_compute_dtype()))
