"""Voxelization: normalise point coordinates into the R^3 grid and average features per voxel.

Reference: modules/voxelization.py:9-28.  Semantics kept exactly:
  c   = coords.detach() - mean over the N points
  normalize=True : c / (max_N ||c||_2 * 2 + eps) + 0.5     (cloud fits the unit cube)
  normalize=False: (c + 1) / 2                             (coords already in the unit ball)
  norm_coords = clamp(c * R, 0, R - 1)      float, returned for the later devoxelization
  vox_coords  = round(norm_coords) -> int32 (round-half-to-even, torch.round)
On the GPU the whole pre-pass is one kernel (csrc/voxelize.hip: voxel_coords_kernel): the per-element
expressions are the reference's fp32 expressions, the mean is the correctly rounded one (fp64 sum) and the
max is exact -- where the reference's two reductions depend on the library's reduction tree.  The torch
formulation below remains for CPU tensors (oracle stack, reference comparison).
"""
import torch
import torch.nn as nn

from . import functional as F
from .functional._autograd import native

__all__ = ['Voxelization']


class Voxelization(nn.Module):
    def __init__(self, resolution, normalize=True, eps=0):
        super().__init__()
        self.r = int(resolution)
        self.normalize = normalize
        self.eps = eps

    def normalized_coords(self, coords):
        """coords (B,3,N) -> float grid coordinates in [0, R-1]; no gradient flows through."""
        centred = coords.detach()
        centred = centred - centred.mean(2, keepdim=True)
        if self.normalize:
            radius = centred.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values
            unit = centred / (radius * 2.0 + self.eps) + 0.5
        else:
            unit = (centred + 1) / 2.0
        return torch.clamp(unit * self.r, 0, self.r - 1)

    def forward(self, features, coords):
        be = native()
        if coords.is_cuda and coords.dtype == torch.float32 and getattr(be, 'has_voxel_coords', False):
            # one launch instead of a dozen tiny library kernels (csrc/voxelize.hip: voxel_coords_kernel)
            norm_coords, vox_coords = be.voxel_coords(coords.detach().contiguous(), self.r, self.normalize, self.eps)
        else:
            norm_coords = self.normalized_coords(coords)
            vox_coords = torch.round(norm_coords).to(torch.int32)
        return F.avg_voxelize(features, vox_coords, self.r), norm_coords

    def extra_repr(self):
        tail = f', normalized eps = {self.eps}' if self.normalize else ''
        return f'resolution={self.r}{tail}'
