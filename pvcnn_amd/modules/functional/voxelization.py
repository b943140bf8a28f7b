"""avg_voxelize: per-voxel mean of point features (reference: modules/functional/voxelization.py:8-40)."""
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['avg_voxelize']


class AvgVoxelization(Function):
    """features (B,C,N) float, coords (B,3,N) integer voxel coordinates, resolution R
    -> (B,C,R,R,R) float.  Saves (ind (B,N), cnt (B,R^3)) for the backward gather."""

    @staticmethod
    @amp_fwd
    def forward(ctx, features, coords, resolution):
        r = int(resolution)
        feats = features.contiguous()
        vox = coords.int().contiguous()
        grid, point_voxel, voxel_count = native().avg_voxelize_forward(feats, vox, r)
        ctx.save_for_backward(point_voxel, voxel_count)
        return grid.view(feats.shape[0], feats.shape[1], r, r, r)

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_grid):
        point_voxel, voxel_count = ctx.saved_tensors
        nb, nc = grad_grid.shape[0], grad_grid.shape[1]
        flat = grad_grid.contiguous().view(nb, nc, -1)
        return native().avg_voxelize_backward(flat, point_voxel, voxel_count), None, None


avg_voxelize = AvgVoxelization.apply
