"""trilinear_devoxelize: 8-corner interpolation of a voxel grid at point positions
(reference: modules/functional/devoxelization.py:8-42).  Note the argument order of the
public function (features, coords, resolution, is_training) versus the native entry point
(resolution, is_training, coords, features) -- both are the reference's."""
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['trilinear_devoxelize']


class TrilinearDevoxelization(Function):
    """features (B,C,R,R,R), coords (B,3,N) float in [0,R-1] -> (B,C,N).  In training mode the
    corner indices/weights (B,8,N) are kept for the backward scatter; in eval mode nothing is
    saved (the native call returns 1-element dummies) and the op is not differentiable,
    exactly like the reference."""

    @staticmethod
    @amp_fwd
    def forward(ctx, features, coords, resolution, is_training=True):
        nb, nc = features.shape[0], features.shape[1]
        grid = features.contiguous().view(nb, nc, -1)
        outs, corner_idx, corner_w = native().trilinear_devoxelize_forward(
            int(resolution), bool(is_training), coords.contiguous(), grid)
        if is_training:
            ctx.save_for_backward(corner_idx, corner_w)
            ctx.r = int(resolution)
        return outs

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_points):
        corner_idx, corner_w = ctx.saved_tensors
        r = ctx.r
        grad_grid = native().trilinear_devoxelize_backward(grad_points.contiguous(), corner_idx, corner_w, r)
        return grad_grid.view(grad_points.shape[0], grad_points.shape[1], r, r, r), None, None, None


trilinear_devoxelize = TrilinearDevoxelization.apply
