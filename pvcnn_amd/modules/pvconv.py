"""PVConv: point-voxel convolution (reference: modules/pvconv.py:11-39).

    (features (B,Cin,N), coords (B,3,N))
        -> devoxelize( voxel_layers( voxelize(features, coords) ) ) + point_features(features)

Sub-module names are the reference's, so released checkpoints load unchanged:
  voxelization, voxel_layers.{0..5}[.6 = SE3d], point_features.layers.{0,1,2}.
The voxel branch's scatter / gather run on the hand-written gfx950 kernels
(avg_voxelize -> pvcnn_avg_voxelize_fwd, trilinear_devoxelize -> pvcnn_trilinear_devox_fwd).
"""
import torch
import torch.nn as nn

from . import functional as F
from .functional._autograd import native
from .functional.bnact import batch_norm_act_devoxelize, batch_norm_act_se_devoxelize, fusable_tail, run_layers
from .functional.conv3d import conv_nsplit, voxel_conv3d
from .se import SE3d
from .shared_mlp import SharedMLP
from .voxelization import Voxelization

__all__ = ['PVConv']


class _VoxelConv3d(nn.Conv3d):
    """nn.Conv3d (same parameters, same state_dict keys) whose 3x3x3 / stride 1 / padding 1 case on the GPU runs this package's
    implicit-GEMM kernels (functional/conv3d.py: f16x2 on the fp16 matrix cores by default) instead of the vendor library."""

    def _fast(self, x):
        """The 3x3x3 / stride 1 / padding 1 cube case this package's implicit-GEMM kernels serve (fp32, or bf16 / fp16 under autocast)."""
        return (x.is_cuda and getattr(native(), 'has_conv3d', False) and self.kernel_size == (3, 3, 3)
                and self.stride == (1, 1, 1) and self.padding == (1, 1, 1) and self.dilation == (1, 1, 1)
                and self.groups == 1 and self.padding_mode == 'zeros' and x.dim() == 5
                and x.shape[2] == x.shape[3] == x.shape[4] and self.weight.dtype == torch.float32
                and (x.dtype == torch.float32 or (torch.is_autocast_enabled() and x.dtype in (torch.bfloat16, torch.float16))))

    def forward(self, x):
        if not self._fast(x):
            return super().forward(x)            # other dtypes / shapes: the vendor library
        return voxel_conv3d(x, self.weight, self.bias, False, conv_nsplit())

    def forward_with_stats(self, x):
        """-> (y, stats_part) on the fast path: the epilogue also emits the per-channel partial sums the
        BatchNorm behind this convolution needs (functional.bnact.run_layers); else plain forward(x)."""
        if not self._fast(x):
            return self.forward(x)
        return voxel_conv3d(x, self.weight, self.bias, True, conv_nsplit())


class PVConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, resolution, with_se=False, normalize=True, eps=0):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.resolution = resolution

        self.voxelization = Voxelization(resolution, normalize=normalize, eps=eps)
        pad = kernel_size // 2
        grid_ops = []
        for cin in (in_channels, out_channels):
            grid_ops += [_VoxelConv3d(cin, out_channels, kernel_size, stride=1, padding=pad),
                         nn.BatchNorm3d(out_channels, eps=1e-4),
                         nn.LeakyReLU(0.1, True)]
        if with_se:
            grid_ops.append(SE3d(out_channels))
        self.voxel_layers = nn.Sequential(*grid_ops)
        self.point_features = SharedMLP(in_channels, out_channels)

    def forward(self, inputs):
        features, coords = inputs
        grid, grid_coords = self.voxelization(features, coords)
        tail = fusable_tail(self.voxel_layers, grid) if self.resolution ** 3 * 4 <= 160 * 1024 else None
        if tail is not None:
            # the last BatchNorm3d + LeakyReLU ride on the devoxelize gather: the activated grid is never written
            # (and the gather does not start on a grid whose write is still draining to HBM: 1.5x slower, see
            # tools/devox_after_writer.py)
            bn, slope, se = tail
            ntail = 2 if se is None else 3
            grid, stats_part = run_layers(self.voxel_layers, grid, stop=len(self.voxel_layers) - ntail, tail_stats=True)
            per_point = self.point_features(features)
            # ... and so does the sum with the point branch: added in the gather's store (one rounded addition, like the
            # reference's `voxel_features + point_features`, modules/pvconv.py:38)
            if se is None:
                fused = batch_norm_act_devoxelize(grid, grid_coords, bn, slope, self.resolution, self.training, stats_part,
                                                  addend=per_point)
            else:   # ... and SE3d: squeeze from one reduction pass over the convolution's output, excitation in the gather's staging
                fused = batch_norm_act_se_devoxelize(grid, grid_coords, bn, slope, se, self.resolution, self.training, stats_part,
                                                     addend=per_point)
            return fused, coords
        else:
            # = self.voxel_layers(grid) with BN + LeakyReLU fused (a hooked Sequential goes through its own __call__)
            hooked = bool(self.voxel_layers._forward_hooks or self.voxel_layers._forward_pre_hooks or self.voxel_layers._backward_hooks)
            grid = self.voxel_layers(grid) if hooked else run_layers(self.voxel_layers, grid)
            per_point = self.point_features(features)
            from_voxels = F.trilinear_devoxelize(grid, grid_coords, self.resolution, self.training)
        return from_voxels + per_point, coords
