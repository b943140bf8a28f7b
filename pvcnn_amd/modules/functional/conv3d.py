"""voxel_conv3d: the 3x3x3, stride-1, padding-1 convolution of PVConv.voxel_layers.

The reference calls nn.Conv3d (cuDNN) here (modules/pvconv.py:20-27).  On gfx950 the three GEMMs
(forward, backward-data, backward-weight) run on the fp32-MFMA implicit-GEMM kernels of
csrc/conv3d.hip; gradients w.r.t. the bias are a plain reduction."""
import torch
from torch.autograd import Function

from . import _cache
from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['voxel_conv3d', 'conv_nsplit', 'bnact_voxel_conv3d']


def conv_nsplit():
    """Arithmetic of the forward / backward-data products, decided where the convolution is CALLED (inside the autograd
    function autocast is already switched off): 1 = bf16 operands under torch.autocast(bfloat16) (BASELINE configs[4]),
    2 = f16x2 (scaled fp16 hi + lo split: fp32-class accuracy at 3 MFMAs per k-step, the default), 3 = bf16x3 split
    (fp32-class, 6 MFMAs), 0 = exact fp32 MFMA."""
    be = native()
    if not getattr(be, 'has_conv3d_split', False):
        return 0
    if torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16:
        return 1
    return getattr(be, 'CONV_NSPLIT', {}).get(getattr(be, 'conv_math', 'fp32'), 0)


class VoxelConv3d(Function):
    @staticmethod
    @amp_fwd
    def forward(ctx, x, weight, bias, want_stats=False, nsplit=0):
        x = x.contiguous()
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.nsplit = int(nsplit)
        b = bias.contiguous() if bias is not None else None
        be = native()
        # f16x2: the input's max |x| (its power-of-two scale) is measured once and reused by backward-weight
        ctx.x_amax = be.absmax_bits(x) if ctx.nsplit == 2 else None
        kw = {'amax': ctx.x_amax} if ctx.nsplit == 2 else {}
        if want_stats:   # second output: BatchNorm partial sums from the epilogue (not differentiable)
            y, part = (be.conv3d_forward_split(x, weight, b, ctx.nsplit, want_stats=True, **kw) if ctx.nsplit
                       else be.conv3d_forward(x, weight, b, want_stats=True))
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)     # no zero tensor for the (non-existent) gradient of `part`
            return y, part
        return be.conv3d_forward_split(x, weight, b, ctx.nsplit, **kw) if ctx.nsplit else be.conv3d_forward(x, weight, b)

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_y, grad_part=None):
        x, weight = ctx.saved_tensors
        if grad_y is None:
            return None, None, None, None, None
        received, grad_y = grad_y, grad_y.contiguous()
        be = native()
        f16 = ctx.nsplit == 2
        wgrad_f16 = f16 and ctx.needs_input_grad[1] and be.conv3d_backward_weight_f16_serves(x)
        # shared by both products; the BatchNorm backward that produced grad_y usually left it on the tensor (_cache.tag_absmax)
        g_amax = _cache.absmax_of(received, lambda: be.absmax_bits(grad_y)) if f16 and (ctx.needs_input_grad[0] or wgrad_f16) else None
        gx = None
        if ctx.needs_input_grad[0]:
            gx = (be.conv3d_backward_data_split(grad_y, weight, ctx.nsplit, **({'amax': g_amax} if f16 else {})) if ctx.nsplit
                  else be.conv3d_backward_data(grad_y, weight))
        want_bias = ctx.has_bias and ctx.needs_input_grad[2]
        gw = gb = None
        if ctx.needs_input_grad[1]:
            # the bias gradient is accumulated by the same kernel from the grad_y tiles it stages anyway
            res = (be.conv3d_backward_weight_f16(x, grad_y, ctx.x_amax, g_amax, with_bias=want_bias) if wgrad_f16
                   else be.conv3d_backward_weight(x, grad_y, with_bias=want_bias))
            gw, gb = res if want_bias else (res, None)
        elif want_bias:
            gb = grad_y.sum(dim=(0, 2, 3, 4))
        return gx, gw, gb, None, None


voxel_conv3d = VoxelConv3d.apply


class BnActVoxelConv3d(Function):
    """conv3d(leaky_relu(batch_norm(x)), weight) + bias -- voxel_layers[1..3] of PVConv (modules/pvconv.py:20-27) as ONE node: the
    BatchNorm3d + LeakyReLU between the two convolutions is applied by the second convolution while it stages its input (forward
    and backward-weight), so the activated grid is never written or read back.  x is the first convolution's raw
    output; its batch statistics come from that convolution's epilogue (stats_part).  f16x2 arithmetic; bit-identical to the
    unfused BatchNormAct -> VoxelConv3d pair."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, bn_weight, bn_bias, running_mean, running_var, use_batch_stats, momentum, eps, slope, stats_part, stats_shift,
                weight, bias, want_stats=False):
        be = native()
        x = x.contiguous()
        x3 = x.view(x.shape[0], x.shape[1], -1)
        weight = weight.contiguous()
        g = bn_weight.contiguous() if bn_weight is not None else None
        b = bn_bias.contiguous() if bn_bias is not None else None
        if use_batch_stats and stats_part is not None:
            mean, rstd = be.bn_finalize(stats_part, x3.shape[0] * x3.shape[2], running_mean, running_var, momentum, eps, stats_shift)
        elif use_batch_stats:
            mean, rstd = be.bn_stats(x3, running_mean, running_var, momentum, eps)
        else:
            mean, rstd = running_mean.clone(), torch.rsqrt(running_var + eps)     # saved for backward: must not alias the live buffers
        bn = (g, b, mean, rstd, float(slope))
        amax = be.bnact_absmax_bits(x, bn)                       # max |act(bn(x))|: one read of x (the activation is never written)
        cb = bias.contiguous() if bias is not None else None
        ctx.save_for_backward(x, g, b, mean, rstd, weight, amax)
        ctx.slope, ctx.use_batch_stats, ctx.has_bias = float(slope), bool(use_batch_stats), bias is not None
        if want_stats:
            y, part = be.conv3d_forward_split_bnact(x, weight, cb, bn, want_stats=True, amax=amax)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y, part
        return be.conv3d_forward_split_bnact(x, weight, cb, bn, amax=amax)

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_y, grad_part=None):
        none = (None,) * 14
        if grad_y is None:
            return none
        x, g, b, mean, rstd, weight, x_amax = ctx.saved_tensors
        be = native()
        received, grad_y = grad_y, grad_y.contiguous()
        g_amax = _cache.absmax_of(received, lambda: be.absmax_bits(grad_y))
        bn = (g, b, mean, rstd, ctx.slope)
        gx = gg = gb = gw = gcb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            g_act = be.conv3d_backward_data_split(grad_y, weight, 2, amax=g_amax)          # gradient w.r.t. the activated grid
            x3 = x.view(x.shape[0], x.shape[1], -1)
            ga3 = g_act.view(x3.shape)
            if getattr(be, 'has_bnact_bwd_absmax', False):
                gx3, gg, gb, amax = be.bnact_backward(x3, ga3, g, b, mean, rstd, ctx.slope, ctx.use_batch_stats, want_amax=True)
                gx = _cache.tag_absmax(gx3.view(x.shape), amax)
            else:
                gx3, gg, gb = be.bnact_backward(x3, ga3, g, b, mean, rstd, ctx.slope, ctx.use_batch_stats)
                gx = gx3.view(x.shape)
        want_bias = ctx.has_bias and ctx.needs_input_grad[12]
        if ctx.needs_input_grad[11]:
            res = be.conv3d_backward_weight_f16_bnact(x, grad_y, x_amax, g_amax, bn, with_bias=want_bias)
            gw, gcb = res if want_bias else (res, None)
        elif want_bias:
            gcb = grad_y.sum(dim=(0, 2, 3, 4))
        return (gx, gg if g is not None else None, gb if b is not None else None, None, None, None, None, None, None, None, None,
                gw, gcb, None)


bnact_voxel_conv3d = BnActVoxelConv3d.apply
