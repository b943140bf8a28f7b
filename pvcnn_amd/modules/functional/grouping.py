"""grouping: gather point features by neighbour index (reference: modules/functional/grouping.py:8-31)."""
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['grouping']


class Grouping(Function):
    """features (B,C,N), indices (B,M,U) int -> (B,C,M,U); backward scatter-adds into (B,C,N)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, features, indices):
        feats = features.contiguous()
        idx = indices.contiguous()
        ctx.save_for_backward(idx)
        ctx.num_points = feats.size(-1)
        return native().grouping_forward(feats, idx)

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_grouped):
        idx, = ctx.saved_tensors
        return native().grouping_backward(grad_grouped.contiguous(), idx, ctx.num_points), None


grouping = Grouping.apply
