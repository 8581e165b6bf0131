"""MI355X drop-in for vq-video-diffusion/autoencoder.py (conv3x3/conv1x1 :8-15, Residual :18-42, ResidualStack
:45-57, SimpleResidualEncoder :60-86, UpscaleResidual :89-131, SimpleResidualDecoder :134-152).

Same classes, constructor signatures, attribute names and state_dict keys (the parameter-holding sub-modules are
ordinary nn.Conv2d / nn.BatchNorm2d, created in the reference's order so that seeded initialisation matches).  The
forward passes run NHWC implicit-GEMM convolutions on MFMA with BatchNorm / LeakyReLU / skip adds fused into the
epilogues (eval mode) or applied by one elementwise kernel after the batch statistics are known (training mode,
which is what the frozen AE runs in inside main.py -- quirk Q3: running statistics are updated even under no_grad).
Tensors cross the module boundary as logical NCHW in channels_last memory format: no layout copies inside; public
`forward()`s return the caller's dtype (boundary dtype rule, see local_3d_attention.py), `forward_nhwc` the compute dtype.
With gradients enabled (VQ-AE training, train_vqae.py:125-192) the blocks run op by op through autograd Functions over
the HIP kernels: data gradient = the same implicit-GEMM conv on the (zero-dilated for stride 2) output gradient with
flipped, transposed weights; weight gradient = split-M MFMA GEMM with an implicit-im2col operand; training-mode
BatchNorm + LeakyReLU backward in two passes; bilinear x2 adjoint in gather form.
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


def _w_build(w):
    co, ci, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if _pad8(ci) != ci or _pad8(co) != co:
        w = F.pad(w, (0, _pad8(ci) - ci, 0, 0, 0, 0, 0, _pad8(co) - co))
    return w.reshape(_pad8(co), -1)


def _w_op(conv, dtype):
    """nn.Conv2d weight [Cout,Cin,KH,KW] -> GEMM operand [Cout8, KH*KW*Cin8] (tap-major, channel fastest; input AND output channels
    zero-padded to multiples of 8: the decoder's 3-channel last layer runs on the direct kernels like every other one, its five
    padding channels are zeros that the caller slices off)."""
    return _cast.operand((conv.weight,), dtype, 'conv', _w_build)


def _to_nhwc(x, dtype):
    """logical NCHW (any memory format) -> contiguous [B,H,W,C8] in the compute dtype."""
    if not x.is_cuda:
        raise WmzError('the conv encoder/decoder runs on the GPU only (no CPU fallback)')
    if (x.dim() == 4 and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16) and not (torch.is_grad_enabled() and x.requires_grad)):
        return ops.nchw_to_nhwc8(x, dtype)                         # flip + channel pad + cast in one launch
    x = x.permute(0, 2, 3, 1)
    c = x.shape[-1]
    if _pad8(c) != c:
        x = F.pad(x, (0, _pad8(c) - c))
    return x.to(dtype).contiguous()


def _to_nchw_view(y):
    """[B,H,W,C] contiguous -> logical NCHW view (channels_last memory format, no copy)."""
    return y.permute(0, 3, 1, 2)


def _grad_path(x, *mods):
    """True when this forward must be differentiable (training through the HIP autograd Functions): some parameter of the
    block is trainable, or the INPUT carries a gradient (a frozen block inside a trainable encoder must still pass the
    gradient on to the layers in front of it)."""
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for m in mods for p in m.parameters()))


def _wT_op(weight, dtype):
    """Operand of the data-gradient conv: w'[ci, kh', kw', co] = w[co, ci, K-1-kh', K-1-kw'] as [Ci8, K*K*Co8]."""
    def build(w):
        co, ci, kh, kw = w.shape
        wt = w.flip(2, 3).permute(1, 2, 3, 0)                     # [ci, kh', kw', co]
        wt = F.pad(wt, (0, _pad8(co) - co, 0, 0, 0, 0, 0, _pad8(ci) - ci))
        return wt.reshape(_pad8(ci), -1)
    return _cast.operand((weight,), dtype, 'convT', build)


FUSE_SKIP_GRAD = True      # A/B (tools/time_vqae_modes.py): False = autograd sums the two gradients of a block's input by an add pass


class _Conv2dFn(torch.autograd.Function):
    """y = conv(x) [+ bias] [+ residual] on the HIP kernels; want_stats: also the per-channel (sum, sum of squares) of the stored
    y -- the batch statistics of the BatchNorm behind the conv -- from the conv's own epilogue instead of a pass over y
    (non-differentiable outputs).  Returns (y, s, q, x'); y has the output channels padded to a multiple of 8.
    passthrough: x' is x handed on for its OTHER consumer (the skip path of a residual block) -- the gradient that consumer sends
    back arrives HERE, as a second output's gradient, and rides in the data-gradient convolution's epilogue (its residual
    operand) instead of an add pass of autograd's behind it."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, residual, want_stats, passthrough=False):
        k = weight.shape[2]
        dt = x.dtype
        bop = None
        if bias is not None:
            bop = bias.detach()
            if _pad8(bop.numel()) != bop.numel():
                bop = F.pad(bop, (0, _pad8(bop.numel()) - bop.numel()))
        out = ops.conv2d_nhwc(x, _cast.operand((weight,), dt, 'conv', _w_build), k, k, stride, pad, bias=bop,
                              residual=None if residual is None else residual.detach(), stats=want_stats)
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        ctx.geom = (k, stride, pad, bias is not None, residual is not None)
        ctx.set_materialize_grads(False)          # (the statistics get no gradient: without this autograd zero-fills two tensors per call)
        xp = x if passthrough else None
        if want_stats:
            y, s, q = out
            ctx.mark_non_differentiable(s, q)
            return y, s, q, xp
        return out, None, None, xp

    @staticmethod
    def backward(ctx, dy, _ds, _dq, dskip):
        x, weight = ctx.saved_tensors
        k, stride, pad, has_bias, has_res = ctx.geom
        if dy is None:
            return (dskip,) + (None,) * 7
        co, ci = weight.shape[:2]
        cop, cip = _pad8(co), x.shape[-1]
        dy = dy.contiguous()
        if dy.shape[-1] != cop:
            dy = F.pad(dy, (0, cop - dy.shape[-1]))
        gw = getattr(weight, '_wmz_grad', None)
        gb = getattr(ctx.bias, '_wmz_grad', None) if has_bias else None
        if gw is not None and (not has_bias or gb is not None):
            # FlatArena (VqaeTrainer): the reduction kernel adds straight into the parameters' gradient slots, in nn.Conv2d's own
            # layout -- no [Cout8, K] temporary, no permute copy, no autograd accumulation pass
            ops.conv2d_nhwc_wgrad(x, dy, k, k, stride, pad, has_bias, into=(gw, gb))
            for prm in (weight, ctx.bias):
                ready = getattr(prm, '_wmz_ready', None) if prm is not None else None
                if ready is not None:
                    ready()
            dw = db = None
        else:
            dwf, dbf = ops.conv2d_nhwc_wgrad(x, dy, k, k, stride, pad, has_bias)
            dw = dwf.view(cop, k, k, cip)[:co, :, :, :ci].permute(0, 3, 1, 2).contiguous()
            db = dbf[:co].contiguous() if has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            B, Hi, Wi, _ = x.shape
            Ho, Wo = dy.shape[1:3]
            if stride == 1:
                dz = dy
            else:
                op_h = Hi - ((Ho - 1) * stride - 2 * pad + k)
                op_w = Wi - ((Wo - 1) * stride - 2 * pad + k)
                dz = ops.dilate_nhwc(dy, (Ho - 1) * stride + 1 + op_h, (Wo - 1) * stride + 1 + op_w, stride)     # zero-inserted dy
            dx = ops.conv2d_nhwc(dz, _wT_op(weight, dy.dtype), k, k, 1, k - 1 - pad,
                                 residual=None if dskip is None else dskip.contiguous())
        elif dskip is not None:
            dx = dskip
        return dx, dw, db, None, None, (dy if has_res else None), None, None


class _BnActFn(torch.autograd.Function):
    """y = act(BatchNorm_train(x) [+ r]); updates the module's running statistics like nn.BatchNorm2d."""

    @staticmethod
    def forward(ctx, x, gamma, beta, r, bn, leaky, s=None, q=None, passthrough=False):
        if s is None:
            s, q = ops.channel_stats_nhwc(x)
        lz = ops.bn_lazy(bn, s, q, _count(x), want_stats=True)      # (finalised by the launch that applies it)
        y = ops.affine_act_nhwc(x, lz, None, b=r, leaky=leaky, slope=LEAKY)
        scale, shift, mean, rstd = lz.scale, lz.shift, lz.mean, lz.rstd
        ctx.leaky, ctx.has_r = leaky, r is not None
        # without a skip input the backward recomputes the LeakyReLU mask from x and (scale, shift): y is not kept for it
        ctx.remask = leaky and r is None and ops.bn_leaky_bwd_supported(x)
        if ctx.remask:
            ctx.save_for_backward(x, scale, shift, mean, rstd, gamma)
        else:
            ctx.save_for_backward(x, y, mean, rstd, gamma)
        ctx.beta = beta
        ctx.set_materialize_grads(False)
        return (y, x) if passthrough else y                  # (x handed on for its other consumer: see _Conv2dFn)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        if dy is None:
            return (dskip,) + (None,) * 8
        if ctx.remask:
            x, scale, shift, mean, rstd, gamma = ctx.saved_tensors
            y, remask = None, (scale, shift)
        else:
            x, y, mean, rstd, gamma = ctx.saved_tensors
            remask = None
        beta = ctx.beta
        gg, gb = getattr(gamma, '_wmz_grad', None), getattr(beta, '_wmz_grad', None)
        if gg is not None and gb is not None and getattr(gamma, '_wmz_single_use', False):
            # VqaeTrainer's arena: zeroed at the start of the step, this layer its slots' only writer -- the reduction's
            # atomics land there (28 zero fills and 28 `grad += g` launches fewer per step)
            dx, _, _, g = ops.bn_act_bwd(x, y, dy, mean, rstd, gamma.detach(), ctx.leaky, LEAKY, into=(gg, gb), remask=remask, add=dskip)
            for prm in (gamma, beta):
                ready = getattr(prm, '_wmz_ready', None)
                if ready is not None:
                    ready()
            return dx, None, None, (g if ctx.has_r else None), None, None, None, None, None
        dx, dgamma, dbeta, g = ops.bn_act_bwd(x, y, dy, mean, rstd, gamma.detach(), ctx.leaky, LEAKY, remask=remask, add=dskip)
        return dx, dgamma, dbeta, (g if ctx.has_r else None), None, None, None, None, None


class _Bilinear2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.bilinear2x_nhwc(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear2x_nhwc_bwd(dy)


def _conv_g(x, conv, residual=None):
    y = _Conv2dFn.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], residual, False)[0]
    # (output channels padded to 8 by the operand; the kernels behind read raw pointers: hand on a contiguous tensor)
    return y if y.shape[-1] == conv.out_channels else y[..., :conv.out_channels].contiguous()


def _conv_bnact_g(x, conv, bn, r=None, leaky=True, passthrough=False):
    """conv -> training-mode BatchNorm (-> + r) -> LeakyReLU under autograd, the batch statistics from the conv's epilogue.
    passthrough: -> (y, x') with x' = x for its other consumer (_Conv2dFn)."""
    if not bn.training or _pad8(conv.out_channels) != conv.out_channels:
        y = _bnact_g(_conv_g(x, conv), bn, r=r, leaky=leaky)
        return (y, x) if passthrough else y
    y, s, q, xp = _Conv2dFn.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], None, True, passthrough)
    y = _BnActFn.apply(y, bn.weight, bn.bias, r, bn, leaky, s, q)
    return (y, xp) if passthrough else y


def _bnact_g(x, bn, r=None, leaky=True, passthrough=False):
    if not bn.training:
        y = _bnact_eval_g(x, bn, r, leaky)
        return (y, x) if passthrough else y
    return _BnActFn.apply(x, bn.weight, bn.bias, r, bn, leaky, None, None, passthrough)


def _bnact_eval_g(x, bn, r, leaky):
    """Eval-mode BatchNorm inside a differentiable forward (an evaluated or frozen auto-encoder that still has to pass gradients on:
    nn.BatchNorm2d in eval mode is the per-channel affine map of its running statistics): y = act(x * scale + shift [+ r]) as torch
    device ops under autograd -- a rare path (the reference trains the VQ-AE in train mode, train_vqae.py:125, and never evaluates
    the frozen one, quirk Q3); the convolutions around it stay on the HIP kernels.  x: NHWC, channels padded to a multiple of 8."""
    c, c8 = bn.num_features, x.shape[-1]
    scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
    shift = bn.bias.float() - bn.running_mean.float() * scale
    if c8 != c:
        scale, shift = F.pad(scale, (0, c8 - c)), F.pad(shift, (0, c8 - c))
    y = x.float() * scale + shift
    if r is not None:
        y = y + r.float()
    if leaky:
        y = F.leaky_relu(y, LEAKY)
    return y.to(x.dtype)


def _conv(x, conv, dtype, **kw):
    k = conv.kernel_size[0]
    bias = conv.bias.detach() if conv.bias is not None else None
    if bias is not None and _pad8(bias.numel()) != bias.numel():
        bias = F.pad(bias, (0, _pad8(bias.numel()) - bias.numel()))
    return ops.conv2d_nhwc(x, _w_op(conv, dtype), k, k, conv.stride[0], conv.padding[0], bias=bias, **kw)


def _count(t):
    return t.numel() // t.shape[-1]


def _check_widths(**widths):
    """The NHWC kernels move channels in groups of 8 (16-byte granules) and the BatchNorm / statistics kernels take the conv's
    padded output as it is: a width behind a BatchNorm that is not a multiple of 8 would be normalised over padding channels.
    The reference's defaults and every published run use 64 / 128 (train_vqae.py:205-206); anything else raises here instead of
    computing something silently different."""
    for name, c in widths.items():
        if c % 8 != 0:
            raise WmzError(f'{name} = {c}: the HIP conv encoder / decoder needs channel widths that are multiples of 8')


class Residual(nn.Module):
    def __init__(self, in_planes, hidden_planes, stride=1, normalize=nn.BatchNorm2d, nonlinearity=nn.LeakyReLU):
        super().__init__()
        _check_widths(in_planes=in_planes, hidden_planes=hidden_planes)
        self._block = nn.Sequential(conv3x3(in_planes, hidden_planes, stride=stride), normalize(hidden_planes),
                                    nonlinearity(inplace=True), conv1x1(hidden_planes, in_planes), normalize(in_planes))
        if stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, in_planes, kernel_size=stride, stride=stride, bias=False),
                                            normalize(in_planes))
        else:
            self.downsample = None

    def forward_nhwc(self, x, dtype):
        c1, bn1, c2, bn2 = self._block[0], self._block[1], self._block[3], self._block[4]
        if _grad_path(x, self):
            # (x has two consumers -- c1 and the skip path; the skip path's gradient comes back through c1's Function and is
            #  summed in its data-gradient convolution's epilogue)
            h, xs = _conv_bnact_g(x, c1, bn1, passthrough=True) if FUSE_SKIP_GRAD else (_conv_bnact_g(x, c1, bn1), x)
            if self.downsample is not None:
                r = _conv_bnact_g(xs, self.downsample[0], self.downsample[1], leaky=False)
            else:
                r = xs
            return _conv_bnact_g(h, c2, bn2, r=r)
        if bn1.training:
            # (every BatchNorm is finalised by the launch that applies it -- ops.bn_lazy: no launch between a conv and its consumer)
            h, s, q = _conv(x, c1, dtype, stats=True)
            lz1 = ops.bn_lazy(bn1, s, q, _count(h))
            # BatchNorm apply + LeakyReLU ride in the 1x1 conv's operand staging (no pass over the hidden tensor)
            h, s, q = _conv(h, c2, dtype, stats=True, pre=(lz1, None, LEAKY))
            lz2 = ops.bn_lazy(bn2, s, q, _count(h))
            if self.downsample is not None:
                r, s, q = _conv(x, self.downsample[0], dtype, stats=True)
                lz3 = ops.bn_lazy(self.downsample[1], s, q, _count(r))
                return ops.affine_act_nhwc(h, lz2, None, r, lz3, None, leaky=True, slope=LEAKY)
            return ops.affine_act_nhwc(h, lz2, None, x, leaky=True, slope=LEAKY)
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
        dt = get_compute_dtype()
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x, dt), dt)).to(x.dtype)


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
        dt = get_compute_dtype()
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x, dt), dt)).to(x.dtype)


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
        dt = get_compute_dtype()
        with ops.stat_arena():                                     # (one zero fill for every BatchNorm statistic of the pass)
            if _grad_path(x, self):
                h = F.leaky_relu(_conv_g(_to_nhwc(x, dt), self._conv_1), LEAKY)
            else:
                h = _conv(_to_nhwc(x, dt), self._conv_1, dt, leaky=True, slope=LEAKY)
            return self._residual_stack.forward_nhwc(h, dt)

    def forward(self, x):
        return _to_nchw_view(self.forward_nhwc(x)).to(x.dtype)


class UpscaleResidual(nn.Module):
    def __init__(self, in_planes, out_planes, upsample):
        super().__init__()
        _check_widths(in_planes=in_planes, out_planes=out_planes)
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
        if _grad_path(x, self):
            # (the skip path's gradient is summed by bn1's backward pass)
            h, x = _bnact_g(x, self.bn1, passthrough=True) if FUSE_SKIP_GRAD else (_bnact_g(x, self.bn1), x)
            if self.upsample:
                h = _Bilinear2xFn.apply(h)
                x = _Bilinear2xFn.apply(x)
            h = _conv_bnact_g(h, self.conv1, self.bn2)
            if self.learn_conv_residual:
                x = _conv_g(x, self.conv_residual)
            return _conv_g(h, self.conv2, residual=x)              # (the skip add rides in conv2's epilogue)
        if self.bn1.training:
            s, q = ops.channel_stats_nhwc(x)
            sc, sh = ops.bn_lazy(self.bn1, s, q, _count(x)), None
        else:
            sc, sh = ops.bn_finalize(self.bn1, None, None, 0)
        h = ops.affine_act_nhwc(x, sc, sh, leaky=True, slope=LEAKY)
        if self.upsample:
            h = ops.bilinear2x_nhwc(h)
            x = ops.bilinear2x_nhwc(x)
        if self.bn2.training:
            h, s, q = _conv(h, self.conv1, dtype, stats=True)
            h = ops.affine_act_nhwc(h, ops.bn_lazy(self.bn2, s, q, _count(h)), None, leaky=True, slope=LEAKY)
        else:
            sc, sh = ops.bn_finalize(self.bn2, None, None, 0)
            h = _conv(h, self.conv1, dtype, scale=sc, shift=sh, leaky=True, slope=LEAKY)
        if self.learn_conv_residual:
            x = _conv(x, self.conv_residual, dtype)
        return _conv(h, self.conv2, dtype, residual=x)

    def forward(self, x):
        dt = get_compute_dtype()
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x, dt), dt)).to(x.dtype)


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

    def forward_nhwc(self, h, raw=False):
        """[B,h,w,E] latents (NHWC, C % 8 == 0) -> logical NCHW image (raw: the last conv's NHWC output as it is, its channels
        padded to a multiple of 8 -- what ops.recon_loss reads)."""
        dt = get_compute_dtype()
        mods = list(self.decoder_stack)
        grad = _grad_path(h, self)
        with ops.stat_arena():
            h = _conv_g(h, mods[0]) if grad else _conv(h, mods[0], dt)
            for m in mods[1:-1]:
                h = m.forward_nhwc(h, dt)
            if raw:
                # (the padded output as the kernel wrote it: no slice copy here, no zero-padding launch in its backward)
                last = mods[-1]
                if grad:
                    return _Conv2dFn.apply(h, last.weight, last.bias, last.stride[0], last.padding[0], None, False)[0]
                return _conv(h, last, dt)
            y = _conv_g(h, mods[-1]) if grad else _conv(h, mods[-1], dt)
            return _to_nchw_view(y)[:, :mods[-1].out_channels]      # (the operand's output channels are padded to 8)

    def forward(self, x):
        return self.forward_nhwc(_to_nhwc(x, get_compute_dtype())).to(x.dtype)
