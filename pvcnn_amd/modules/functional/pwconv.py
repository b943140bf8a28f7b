"""pointwise_conv: the kernel-size-1 Conv1d / Conv2d of SharedMLP (reference: modules/shared_mlp.py:9-25).

The reference calls nn.Conv1d / nn.Conv2d (cuDNN / cuBLAS).  On gfx950 the three GEMMs (forward, backward-data,
backward-weight + bias gradient) run directly on the channel-major (B, C, N) tensors in "f16x2" arithmetic on the fp16 matrix
cores (csrc/pointwise_bf16.hip, pointwise_wgrad_f16.hip: fp32 tensors, operands split into scaled fp16 hi + lo, fp32 accumulation)
-- forward / backward-data from `backend.pw_split_min_macs` (16.8 M) multiply-adds up, backward-weight from
`backend.pw_wgrad_f16_min_macs` (4.3 G) -- and on the fp32-MFMA kernels of csrc/pointwise.hip below those bars."""
import torch
from torch.autograd import Function

from . import _cache, _gradslots
from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['pointwise_conv', 'pw_nsplit']


def pw_nsplit(x, weight):
    """Arithmetic of one 1x1 convolution's forward / backward-data products, decided where it is CALLED (inside the autograd node
    autocast is already switched off): 2 = f16x2 / 3 = bf16x3 (fp32-class, csrc/pointwise_bf16.hip; `backend.pw_math`), 1 = plain bf16
    operands under torch.autocast(bfloat16), 0 = the fp32-MFMA kernels of csrc/pointwise.hip -- always for GEMMs below
    `backend.pw_split_min_macs` multiply-adds: they are launch-bound, and the split kernels' extra launches cost more than they save."""
    be = native()
    if not getattr(be, 'has_pwconv_split', False):
        return 0
    macs = weight.shape[0] * weight.shape[1] * x.shape[0] * (x.numel() // max(x.shape[0] * x.shape[1], 1))
    if macs < getattr(be, 'pw_split_min_macs', 0):
        return 0
    if torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16:
        return 1
    return getattr(be, 'PW_NSPLIT', {}).get(getattr(be, 'pw_math', 'fp32'), 0)


class PointwiseConv(Function):
    @staticmethod
    @amp_fwd
    def forward(ctx, x, weight, bias, want_stats=False, split=None):
        shape = x.shape
        x3 = x.contiguous().view(shape[0], shape[1], -1)
        w2 = weight.contiguous().view(weight.shape[0], weight.shape[1])
        ctx.save_for_backward(x3, w2)
        ctx.has_bias, ctx.x_shape, ctx.w_shape = bias is not None, shape, weight.shape
        ctx.bias_param = bias                     # (only asked where its gradient should be written: _gradslots.claim)
        b = bias.contiguous() if bias is not None else None
        be = native()
        # the split products on the 16-bit matrix cores (csrc/pointwise_bf16.hip): 2 = f16x2, 3 = bf16x3, 1 = bf16, 0 = fp32 MFMA
        ctx.split = int(split) if split is not None else pw_nsplit(x3, w2)
        ctx.x_amax = None
        if ctx.split in (1, 2):     # the input's amax buffer (one scale per 256-point tile): left on it by its producer, else one read
            ctx.x_amax = _cache.amax_of(x, be.PW_AMAX_SEG)
            if ctx.x_amax is None and ctx.split == 2:             # (bf16 mode: only backward-weight wants it, and measures it itself)
                ctx.x_amax = be.pw_amax(x3, want_global=False)        # (every consumer below takes the table)
        akw = {'amax': ctx.x_amax} if ctx.split == 2 else {}
        # forward + backward-data weight images from one launch when the input wants a gradient (see functional/conv3d.py)
        ctx.w_bwd_image = None
        if ctx.split and hasattr(be, 'pw_weight_images') and ctx.needs_input_grad[0]:
            w_image, ctx.w_bwd_image = be.pw_weight_images(w2, ctx.split)
            run = lambda **kw: be.pwconv_gemm_split(x3, w_image, b, w2.shape[0], ctx.split, amax=ctx.x_amax, **kw)
        elif ctx.split:
            run = lambda **kw: be.pwconv_forward_split(x3, w2, b, ctx.split, **akw, **kw)
        else:
            run = lambda **kw: be.pwconv_forward(x3, w2, b, **kw)
        if want_stats:   # second output: BatchNorm partial sums from the epilogue (not differentiable)
            y, part = run(want_stats=True)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)     # no zero tensor for the (non-existent) gradient of `part`
            return y.view(shape[0], w2.shape[0], *shape[2:]), part
        return run().view(shape[0], w2.shape[0], *shape[2:])

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_y, grad_part=None):
        x3, w2 = ctx.saved_tensors
        if grad_y is None:
            return None, None, None, None, None
        g3 = grad_y.contiguous().view(x3.shape[0], w2.shape[0], -1)
        be = native()
        f16 = ctx.split == 2
        # the f16x2 backward-weight kernel also serves the bf16 (autocast) mode: more accurate than bf16 operands, and far faster than
        # the fp32-MFMA kernel on the large GEMMs this mode is chosen for
        wgrad_f16 = (ctx.split in (1, 2) and ctx.needs_input_grad[1] and be.pwconv_backward_weight_f16_serves(x3)
                     and x3.shape[0] * x3.shape[2] * w2.shape[0] * w2.shape[1] >= getattr(be, 'pw_wgrad_f16_min_macs', 0))
        # shared by both products; the BatchNorm backward that produced grad_y left it on the tensor (_cache.tag_amax)
        g_amax = None
        if (f16 and (ctx.needs_input_grad[0] or wgrad_f16)) or (ctx.split == 1 and wgrad_f16):
            g_amax = _cache.amax_of(grad_y, be.PW_AMAX_SEG)
            if g_amax is None and f16:
                g_amax = be.pw_amax(g3, want_global=False)
        gx = None
        if ctx.needs_input_grad[0]:
            if ctx.split and ctx.w_bwd_image is not None:
                gx = be.pwconv_gemm_split(g3, ctx.w_bwd_image, None, w2.shape[1], ctx.split, False, g_amax).view(ctx.x_shape)
            else:
                gx = (be.pwconv_backward_data_split(g3, w2, ctx.split, **({'amax': g_amax} if f16 else {})) if ctx.split
                      else be.pwconv_backward_data(g3, w2)).view(ctx.x_shape)
        want_bias = ctx.has_bias and ctx.needs_input_grad[2]
        gw = gb = None
        if ctx.needs_input_grad[1]:
            # (x_amax / g_amax None -- the bf16 mode measured neither: backward-weight takes the global maxima in one read each)
            dst = _gradslots.destinations(be, w2, ctx.bias_param if want_bias else None)   # the parameters' slots in a flat gradient bucket
            res = (be.pwconv_backward_weight_f16(x3, g3, ctx.x_amax, g_amax, with_bias=want_bias, **dst) if wgrad_f16
                   else be.pwconv_backward_weight(x3, g3, with_bias=want_bias, **dst))
            gw, gb = res if want_bias else (res, None)
            gw = gw.view(ctx.w_shape)
        elif want_bias:
            gb = g3.sum(dim=(0, 2))
        return gx, gw, gb, None, None


pointwise_conv = PointwiseConv.apply
