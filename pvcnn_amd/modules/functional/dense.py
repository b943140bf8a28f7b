"""linear_bn_relu: nn.Linear + nn.BatchNorm1d + nn.ReLU on (rows, C) with a handful of rows -- the `_linear_bn_relu` blocks of the
reference's heads (models/utils.py:11-12; models/s3dis/pvcnn.py:22-25 applies two of them to ONE row per cloud).

As modules such a block is 5 launches forward and 6 backward for microseconds of arithmetic; in training mode on the GPU it is one
launch forward and two backward here (csrc/dense.hip: the workgroup that owns an output channel owns its whole batch, so the batch
statistics never leave it; grad_x is one library GEMM).  `run_dense(seq, x)` walks an nn.Sequential and fuses every
(Linear, BatchNorm1d, ReLU) triple it can serve; everything else -- eval mode, CPU tensors, autocast to 16 bits, hooked modules, too
many rows -- goes through the modules themselves.  Parameters, buffers and state_dict keys are those of the plain modules."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _gradslots
from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['linear_bn_relu', 'run_dense']


class LinearBnRelu(Function):
    @staticmethod
    @amp_fwd
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, counter, eps, momentum):
        xc, w = x.contiguous(), weight.contiguous()
        y, z, mean, rstd = native().dense_bn_relu_forward(xc, w, bias, gamma, beta, running_mean, running_var, counter, eps, momentum)
        ctx.save_for_backward(xc, w, z, mean, rstd, gamma, beta)
        ctx.params = (weight, bias, gamma, beta)          # (only asked where their gradients should be written: _gradslots.claim)
        return y

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_y):
        if grad_y is None:
            return (None,) * 10
        xc, w, z, mean, rstd, gamma, beta = ctx.saved_tensors
        be = native()
        weight, bias, g_param, b_param = ctx.params
        need = ctx.needs_input_grad
        dst = {}
        if need[1]:
            lin = _gradslots.destinations(be, weight, bias if (bias is not None and need[2]) else None)
            dst.update(lin)
        if gamma is not None and need[3]:
            bn = _gradslots.destinations(be, g_param, b_param if need[4] else None)
            if 'out_w' in bn:
                dst['out_gamma'] = bn['out_w']
            if 'out_b' in bn:
                dst['out_beta'] = bn['out_b']
        gz, gw, gb, gg, gbeta = be.dense_bn_relu_backward(xc, grad_y.contiguous(), z, mean, rstd, gamma, beta, **dst)
        gx = gz @ w if need[0] else None
        return (gx, gw if need[1] else None, gb if (bias is not None and need[2]) else None,
                gg if (gamma is not None and need[3]) else None, gbeta if (beta is not None and need[4]) else None,
                None, None, None, None, None)


# PVCNN_DENSE_HEAD=0 (read once per process): the torch modules for every block (A/B)
_ENABLED = __import__('os').environ.get('PVCNN_DENSE_HEAD', '1') != '0'


def _servable(lin, bn, act, x):
    be = native() if (x.is_cuda and _ENABLED) else None
    return (be is not None and getattr(be, 'has_dense_bn_relu', False) and type(lin) is nn.Linear and type(bn) is nn.BatchNorm1d
            and type(act) is nn.ReLU and bn.training and bn.momentum is not None and x.dim() == 2 and x.dtype == torch.float32
            and lin.weight.dtype == torch.float32 and not torch.is_autocast_enabled()
            and (not bn.track_running_stats or (bn.running_mean is not None and bn.num_batches_tracked is not None
                                                and bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64))
            and bn.affine == (bn.weight is not None)
            and be.dense_bn_relu_supported(x.shape[0], lin.in_features, lin.out_features)
            and not any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in (lin, bn, act)))


def linear_bn_relu(lin, bn, act, x):
    """act(bn(lin(x))) for the module triple; one launch forward on the training GPU path, the modules themselves elsewhere."""
    if not _servable(lin, bn, act, x):
        return act(bn(lin(x)))
    track = bn.track_running_stats
    return LinearBnRelu.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean if track else None,
                              bn.running_var if track else None, bn.num_batches_tracked if track else None, bn.eps, bn.momentum)


def run_dense(seq, x):
    """seq(x) for an nn.Sequential whose (Linear, BatchNorm1d, ReLU) triples -- directly inside it or wrapped in an inner Sequential of
    exactly those three (models/utils.py:11-12) -- are fused where possible; other members are called as they are."""
    if seq._forward_hooks or seq._forward_pre_hooks or seq._backward_hooks:
        return seq(x)
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if type(m) is nn.Sequential and len(m) == 3 and not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks):
            x = linear_bn_relu(m[0], m[1], m[2], x) if (isinstance(m[0], nn.Linear) and torch.is_tensor(x)) else m(x)
            i += 1
        elif i + 2 < len(mods) and isinstance(m, nn.Linear) and isinstance(mods[i + 1], nn.BatchNorm1d) and isinstance(mods[i + 2], nn.ReLU):
            x = linear_bn_relu(m, mods[i + 1], mods[i + 2], x)
            i += 3
        else:
            x = m(x)
            i += 1
    return x
