"""voxel_conv3d: the 3x3x3, stride-1, padding-1 convolution of PVConv.voxel_layers.

The reference calls nn.Conv3d (cuDNN) here (modules/pvconv.py:20-27).  On gfx950 the three GEMMs (forward, backward-data,
backward-weight) run on hand-written implicit-GEMM kernels: by default in "f16x2" arithmetic on the fp16 matrix cores (fp32
tensors, operands split into scaled fp16 hi + lo, fp32 accumulation: csrc/conv3d_bf16.hip, conv3d_wgrad_f16.hip), plain bf16
operands under torch.autocast, exact fp32 MFMA (csrc/conv3d.hip) with PVCNN_CONV_MATH=fp32 and for the grids the f16x2
backward-weight kernel does not serve.  The bias gradient rides on the backward-weight kernel."""
import torch
from torch.autograd import Function

from . import _cache, _gradslots
from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['voxel_conv3d', 'conv_nsplit']


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
        given, x = x, x.contiguous()
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias                     # (only asked where its gradient should be written: _gradslots.claim)
        ctx.nsplit = int(nsplit)
        b = bias.contiguous() if bias is not None else None
        be = native()
        # f16x2: the input's amax buffer (its power-of-two scales, one per z row) -- left on the tensor by the BatchNorm pass that
        # wrote it (_cache.tag_amax), else measured here in one read -- is reused by backward-weight
        ctx.x_amax = None
        if ctx.nsplit in (1, 2):
            ctx.x_amax = _cache.amax_of(given, x.shape[2])
            if ctx.x_amax is None and ctx.nsplit == 2:           # (bf16 mode: only backward-weight wants it, and measures it itself)
                ctx.x_amax = be.conv_amax(x, want_global=False)      # (every consumer below takes the table)
        kw = {'amax': ctx.x_amax} if ctx.nsplit == 2 else {}
        # the pre-split weight images: when the input wants a gradient the backward-data image is made by the SAME launch as the
        # forward one and kept for backward (the values backward must use are the ones saved now, not a later state of the weight)
        ctx.w_bwd_image = None
        if ctx.nsplit and hasattr(be, 'conv_weight_images') and ctx.needs_input_grad[0]:
            w_image, ctx.w_bwd_image = be.conv_weight_images(weight, ctx.nsplit)
            run = lambda **k: be.conv3d_igemm_split(x, w_image, b, weight.shape[0], ctx.nsplit, amax=ctx.x_amax, **k)
        elif ctx.nsplit:
            run = lambda **k: be.conv3d_forward_split(x, weight, b, ctx.nsplit, **kw, **k)
        else:
            run = lambda **k: be.conv3d_forward(x, weight, b, **k)
        if want_stats:   # second output: BatchNorm partial sums from the epilogue (not differentiable)
            y, part = run(want_stats=True)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)     # no zero tensor for the (non-existent) gradient of `part`
            return y, part
        return run()

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_y, grad_part=None):
        x, weight = ctx.saved_tensors
        if grad_y is None:
            return None, None, None, None, None
        received, grad_y = grad_y, grad_y.contiguous()
        be = native()
        f16 = ctx.nsplit == 2
        # the f16x2 backward-weight kernel also serves the bf16 (autocast) mode: more accurate than bf16 operands and 2.6x the rate
        # of the fp32-MFMA kernel (x_amax / g_amax are None there: the kernel's wrapper takes the global maxima in one read each)
        wgrad_f16 = ctx.nsplit in (1, 2) and ctx.needs_input_grad[1] and be.conv3d_backward_weight_f16_serves(x)
        # shared by both products; the BatchNorm backward that produced grad_y left it on the tensor (_cache.tag_amax)
        g_amax = None
        if (f16 and (ctx.needs_input_grad[0] or wgrad_f16)) or (ctx.nsplit == 1 and wgrad_f16):
            g_amax = _cache.amax_of(received, grad_y.shape[2])
            if g_amax is None and f16:
                g_amax = be.conv_amax(grad_y, want_global=False)
        gx = None
        if ctx.needs_input_grad[0]:
            if ctx.nsplit and ctx.w_bwd_image is not None:     # a convolution with Ci and Co exchanged on the flipped weights (forward's image)
                gx = be.conv3d_igemm_split(grad_y, ctx.w_bwd_image, None, weight.shape[1], ctx.nsplit, False, g_amax)
            else:
                gx = (be.conv3d_backward_data_split(grad_y, weight, ctx.nsplit, **({'amax': g_amax} if f16 else {})) if ctx.nsplit
                      else be.conv3d_backward_data(grad_y, weight))
        want_bias = ctx.has_bias and ctx.needs_input_grad[2]
        gw = gb = None
        if ctx.needs_input_grad[1]:
            # the bias gradient is accumulated by the same kernel from the grad_y tiles it stages anyway
            dst = _gradslots.destinations(be, weight, ctx.bias_param if want_bias else None)   # the parameters' slots in a flat gradient bucket
            res = (be.conv3d_backward_weight_f16(x, grad_y, ctx.x_amax, g_amax, with_bias=want_bias, **dst) if wgrad_f16
                   else be.conv3d_backward_weight(x, grad_y, with_bias=want_bias, **dst))
            gw, gb = res if want_bias else (res, None)
        elif want_bias:
            gb = grad_y.sum(dim=(0, 2, 3, 4))
        return gx, gw, gb, None, None


voxel_conv3d = VoxelConv3d.apply
