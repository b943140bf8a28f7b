"""Voxelization: normalise point coordinates into the R^3 grid and average features per voxel.

Reference: modules/voxelization.py:9-28.  Semantics kept exactly:
  c   = coords.detach() - mean over the N points
  normalize=True : c / (max_N ||c||_2 * 2 + eps) + 0.5     (cloud fits the unit cube)
  normalize=False: (c + 1) / 2                             (coords already in the unit ball)
  norm_coords = clamp(c * R, 0, R - 1)      float, returned for the later devoxelization
  vox_coords  = round(norm_coords) -> int32 (round-half-to-even, torch.round)

GPU path (default): the two REDUCTIONS are the reference's own torch calls (`mean(2)`, `norm(dim=1).max(dim=2)`)
on the same device -- same library kernels, same reduction trees -- and everything elementwise behind them is one
fused kernel (csrc/voxelize.hip: voxel_coords_tail_kernel) whose steps are rounded one by one like the reference's
separate kernels: norm_coords and vox_coords are BIT-IDENTICAL to the reference formulation on that device
(tests/test_gpu_voxel_coords.py).  The statistics do not depend on R and the grid coordinates do not depend on the
layer, so both are memoised per coords tensor (functional/_cache.py): the three R = 16 PVConvs of PVCNN share one
pre-pass.  `Voxelization.single_launch = True` opts into the one-launch kernel with order-free statistics (fp64 mean,
exact max: voxel_coords_kernel), which can differ from torch in the last bit of the mean and hence in a voxel id
of a point that sits on a rounding boundary.
"""
import torch
import torch.nn as nn

from . import functional as F
from .functional import _cache
from .functional._autograd import native

__all__ = ['Voxelization']


class Voxelization(nn.Module):
    single_launch = False      # opt-in: order-free statistics in one launch (not bit-identical to torch's reductions)

    def __init__(self, resolution, normalize=True, eps=0):
        super().__init__()
        self.r = int(resolution)
        self.normalize = normalize
        self.eps = eps

    def normalized_coords(self, coords):
        """coords (B,3,N) -> float grid coordinates in [0, R-1]; no gradient flows through.
        The reference formulation op by op (modules/voxelization.py:17-23); used for CPU tensors and as the
        same-device checker of the fused GPU path."""
        centred = coords.detach()
        centred = centred - centred.mean(2, keepdim=True)
        if self.normalize:
            radius = centred.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values
            unit = centred / (radius * 2.0 + self.eps) + 0.5
        else:
            unit = (centred + 1) / 2.0
        return torch.clamp(unit * self.r, 0, self.r - 1)

    @staticmethod
    def _statistics(coords, normalize):
        """(mean (B,3,1), radius (B,1,1) | None): the reference's reductions, called exactly as it calls them."""
        c = coords.detach()
        mean = c.mean(2, keepdim=True)
        radius = None
        if normalize:
            radius = (c - mean).norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values
        return mean, radius

    def grid_coordinates(self, coords):
        """coords (B,3,N) on the GPU -> (norm_coords float (B,3,N), vox_coords int32 (B,3,N))."""
        be = native()
        normalize = bool(self.normalize)
        if self.single_launch:
            return self._fresh(coords, ('vox1', self.r, normalize, float(self.eps)),
                               lambda: be.voxel_coords(coords.detach().contiguous(), self.r, normalize, self.eps))

        def tail():
            mean, radius = _cache.memo(coords, ('stats', normalize), lambda: self._statistics(coords, normalize))
            c = coords.detach()
            from .functional.backend import batch_strided_ok
            if not batch_strided_ok(c):
                c = c.contiguous()
            return be.voxel_coords_tail(c, mean.contiguous(), radius.contiguous() if radius is not None else None, self.r, self.eps)
        return self._fresh(coords, ('vox', self.r, normalize, float(self.eps)), tail)

    @staticmethod
    def _fresh(coords, key, make):
        """memo(coords, key) of a (norm_coords, vox_coords) pair that is handed OUT of this module (forward returns norm_coords): the
        entry also records both tensors' in-place version counters, and a caller that modified one of them in place gets a recomputed
        pair next time instead of its own edit (the other layers at this resolution share the pair)."""
        def stamped():
            norm, vox = make()
            return norm, vox, norm._version, vox._version
        for _ in range(2):
            norm, vox, vn, vv = _cache.memo(coords, key, stamped)
            if norm._version == vn and vox._version == vv:
                return norm, vox
            _cache.forget(coords, key)
        return norm, vox

    def forward(self, features, coords):
        be = native()
        if coords.is_cuda and coords.dtype == torch.float32 and coords.dim() == 3 and getattr(be, 'has_voxel_coords', False):
            norm_coords, vox_coords = self.grid_coordinates(coords)
            # (training mode: the PVConv around this module devoxelizes with is_training = True and its backward scatters -- also when
            #  the features themselves need no gradient: the voxel convolutions' weights do)
            if getattr(be, 'has_pvconv_plans', False) and torch.is_grad_enabled() and self.training:
                self._plan_pair(be, norm_coords, vox_coords)
        else:
            norm_coords = self.normalized_coords(coords)
            vox_coords = torch.round(norm_coords).to(torch.int32)
        return F.avg_voxelize(features, vox_coords, self.r), norm_coords

    def _plan_pair(self, be, norm_coords, vox_coords):
        """Training: the voxelize plan of (vox_coords, R) AND the devoxelize-backward plan of (norm_coords, R) -- the PVConv around this
        module devoxelizes there (modules/pvconv.py:36) and its backward scatters through that plan -- from one launch chain
        (backend.pvconv_plans) the first time a layer voxelizes these coordinates.  Both land where the functional layer looks for
        them: the memo of functional.voxelization.AvgVoxelization and the CornerTaps holder of functional.devoxelization."""
        from .functional import _cache
        from .functional.devoxelization import CornerTaps
        built = []

        def make():
            pair = be.pvconv_plans(vox_coords, norm_coords.contiguous(), self.r) if norm_coords.is_contiguous() and vox_coords.is_contiguous() else None
            if pair is None:
                return be.avg_voxelize_plan(vox_coords.contiguous(), self.r)
            built.append(pair[1])
            return pair[0]
        _cache.memo(vox_coords, ('avg_voxelize_plan', self.r), make)
        if built:
            taps = CornerTaps.of(norm_coords, self.r)
            if taps.plan is None:
                taps.plan = built[0]

    def extra_repr(self):
        tail = f', normalized eps = {self.eps}' if self.normalize else ''
        return f'resolution={self.r}{tail}'
