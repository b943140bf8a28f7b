"""avg_voxelize: per-voxel mean of point features (reference: modules/functional/voxelization.py:8-40)."""
from torch.autograd import Function

from . import _cache
from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['avg_voxelize']


class AvgVoxelization(Function):
    """features (B,C,N) float, coords (B,3,N) integer voxel coordinates, resolution R
    -> (B,C,R,R,R) float.  Saves (ind (B,N), cnt (B,R^3)) for the backward gather.

    On the GPU the scatter is "plan, then apply" (csrc/csr.h): the plan -- the per-cloud counting sort of the points by
    voxel -- depends on (coords, R) only and is memoised per coords tensor, so the layers of a network that voxelize
    the same coordinates at the same resolution (PVCNN: three PVConvs at R = 16) sort once."""

    @staticmethod
    @amp_fwd
    def forward(ctx, features, coords, resolution):
        r = int(resolution)
        feats = features.contiguous()
        be = native()
        plan = None
        if feats.is_cuda and getattr(be, 'has_scatter_plans', False):
            plan = _cache.memo(coords, ('avg_voxelize_plan', r), lambda: be.avg_voxelize_plan(coords.int().contiguous(), r))
        if plan is not None:
            grid, point_voxel, voxel_count = be.avg_voxelize_apply(feats, plan), plan.ind, plan.cnt
        else:
            grid, point_voxel, voxel_count = be.avg_voxelize_forward(feats, coords.int().contiguous(), r)
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
