"""trilinear_devoxelize: 8-corner interpolation of a voxel grid at point positions
(reference: modules/functional/devoxelization.py:8-42).  Note the argument order of the
public function (features, coords, resolution, is_training) versus the native entry point
(resolution, is_training, coords, features) -- both are the reference's."""
from torch.autograd import Function

from . import _cache
from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['trilinear_devoxelize', 'CornerTaps']


class CornerTaps:
    """What a devoxelization derives from (coords, R) alone: the corner indices / weights (B,8,N) the reference saves
    for backward, and the counting-sort plan of the backward scatter.  One holder per (coords tensor, R) is shared by
    every layer that devoxelizes there (functional/_cache.py): the first forward emits inds / wgts, the first backward
    builds the plan, the others reuse both."""
    __slots__ = ('inds', 'wgts', 'plan', 'r')

    def __init__(self, r):
        self.inds = self.wgts = self.plan = None
        self.r = int(r)

    @staticmethod
    def of(coords, r):
        """The shared holder for (coords, r) when the active backend plans its scatters, else a private one."""
        if coords.is_cuda and getattr(native(), 'has_scatter_plans', False):
            return _cache.memo(coords, ('corner_taps', int(r)), lambda: CornerTaps(r))
        return CornerTaps(r)

    def forward(self, run):
        """run(is_training) -> [outs, inds, wgts] of a native forward call; emits inds / wgts only the first time."""
        if self.inds is None:
            outs, self.inds, self.wgts = run(True)
        else:
            outs = run(False)[0]
        return outs

    def backward(self, grad_points):
        """grad (B,C,N) (rows of a cloud contiguous) -> grad grid (B,C,R^3)."""
        be = native()
        if getattr(be, 'has_scatter_plans', False) and grad_points.is_cuda:
            if self.plan is None:
                made = be.trilinear_devoxelize_backward_plan(self.inds, self.wgts, self.r)
                self.plan = made if made is not None else False
            if self.plan is not False:
                return be.trilinear_devoxelize_backward_apply(grad_points, self.plan, self.r)
        return be.trilinear_devoxelize_backward(grad_points, self.inds, self.wgts, self.r)


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
        r = int(resolution)
        if not is_training:
            return native().trilinear_devoxelize_forward(r, False, coords.contiguous(), grid)[0]
        taps = CornerTaps.of(coords, r)
        pts = coords.contiguous()
        outs = taps.forward(lambda emit: native().trilinear_devoxelize_forward(r, emit, pts, grid))
        ctx.save_for_backward(taps.inds, taps.wgts)
        ctx.taps = taps
        return outs

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_points):
        r = ctx.taps.r
        grad_grid = ctx.taps.backward(grad_points.contiguous())
        return grad_grid.view(grad_points.shape[0], grad_points.shape[1], r, r, r), None, None, None


trilinear_devoxelize = TrilinearDevoxelization.apply
