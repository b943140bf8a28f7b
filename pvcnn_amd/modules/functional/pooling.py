"""neighbor_max: the max over the neighbours of a centre, `x.max(dim=-1).values` on (B, C, M, K) (reference:
modules/pointnet.py:85), as one streaming pass forward and one backward (csrc/pool.hip).  Not one of the reference's ten
`modules.functional` names: PointNetSAModule calls it where the reference calls torch.max; every case the kernel does not cover
(CPU tensors, other dtypes, K not in {4, 8, 16, 32, 64}, no native backend) IS torch.max."""
import torch
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['neighbor_max']


def _aligned(t):
    """contiguous AND on a 16-byte boundary (the kernels read rows with 16-byte loads): a contiguous view at an odd storage offset is
    copied to a fresh allocation instead of being refused by the library."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


class NeighborMax(Function):
    @staticmethod
    @amp_fwd
    def forward(ctx, x):
        x = _aligned(x)
        out, winners = native().neighbor_max_forward(x)
        ctx.save_for_backward(winners)
        ctx.k = x.shape[-1]
        return out

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_out):
        winners, = ctx.saved_tensors
        return native().neighbor_max_backward(_aligned(grad_out), winners, ctx.k)


def neighbor_max(x):
    """x (..., K) -> (...): max over the last dimension (ties: the first index gets the gradient, torch's rule)."""
    be = native() if x.is_cuda else None
    if (be is not None and getattr(be, 'has_neighbor_max', False) and x.dtype == torch.float32 and x.dim() >= 2 and x.numel() > 0
            and be.neighbor_max_supported(x.shape[-1])):
        return NeighborMax.apply(x)
    return x.max(dim=-1).values
