"""PointNet++ building blocks used by PVCNN++ (reference: modules/pointnet.py:11-111).

  PointNetAModule   global abstraction: SharedMLP(s) over all points -> max over N -> one "point"
  PointNetSAModule  set abstraction: FPS centres -> per-scale BallQuery -> SharedMLP(dim=2) -> max over U
  PointNetFPModule  feature propagation: 3-NN interpolation of centre features (+ skip) -> SharedMLP

Attribute names (`mlps`, `groupers`, `mlp`) and `out_channels` bookkeeping follow the reference
so that model builders and checkpoints are interchangeable."""
import torch
import torch.nn as nn

from . import functional as F
from .ball_query import BallQuery
from .functional.pooling import neighbor_max
from .shared_mlp import SharedMLP

__all__ = ['PointNetAModule', 'PointNetSAModule', 'PointNetFPModule']


def _as_nested(widths, copies=1):
    """out_channels spec -> list (one entry per scale) of per-layer width lists."""
    if not isinstance(widths, (list, tuple)):
        return [[widths]] * copies
    if not isinstance(widths[0], (list, tuple)):
        return [widths] * copies
    return widths


class PointNetAModule(nn.Module):
    def __init__(self, in_channels, out_channels, include_coordinates=True):
        super().__init__()
        scales = _as_nested(out_channels)
        cin = in_channels + (3 if include_coordinates else 0)
        self.include_coordinates = include_coordinates
        self.out_channels = sum(s[-1] for s in scales)
        self.mlps = nn.ModuleList([SharedMLP(in_channels=cin, out_channels=s, dim=1) for s in scales])

    def forward(self, inputs):
        features, coords = inputs
        if self.include_coordinates:
            features = torch.cat([features, coords], dim=1)
        origin = torch.zeros((coords.size(0), 3, 1), device=coords.device)
        pooled = [mlp(features).max(dim=-1, keepdim=True).values for mlp in self.mlps]
        return (torch.cat(pooled, dim=1) if len(pooled) > 1 else pooled[0]), origin

    def extra_repr(self):
        return f'out_channels={self.out_channels}, include_coordinates={self.include_coordinates}'


class PointNetSAModule(nn.Module):
    def __init__(self, num_centers, radius, num_neighbors, in_channels, out_channels, include_coordinates=True):
        super().__init__()
        radii = list(radius) if isinstance(radius, (list, tuple)) else [radius]
        ks = list(num_neighbors) if isinstance(num_neighbors, (list, tuple)) else [num_neighbors] * len(radii)
        assert len(radii) == len(ks)
        scales = _as_nested(out_channels, copies=len(radii))
        assert len(radii) == len(scales)
        cin = in_channels + (3 if include_coordinates else 0)
        self.num_centers = num_centers
        self.out_channels = sum(s[-1] for s in scales)
        self.groupers = nn.ModuleList(
            [BallQuery(radius=r, num_neighbors=k, include_coordinates=include_coordinates) for r, k in zip(radii, ks)])
        self.mlps = nn.ModuleList([SharedMLP(in_channels=cin, out_channels=s, dim=2) for s in scales])

    def forward(self, inputs):
        features, coords = inputs
        ahead = self.__dict__.pop('_centers_ahead', None)
        centers = None
        if ahead is not None and coords.is_cuda:
            # (pvcnn_amd.workload.centers_ahead: the sampling of the whole pyramid depends on the input coordinates alone and was
            #  issued on a stream of its own at the top of the network's forward; the SAME indices F.furthest_point_sample computes,
            #  valid for exactly the tensor the level before returned -- anything else samples in line)
            picked, done, chain = ahead
            torch.cuda.current_stream().wait_event(done)      # always: the side path joins here (a capture must not end with it open)
            if chain.accepts(coords):
                picked.record_stream(torch.cuda.current_stream())
                centers = F.gather(coords.contiguous(), picked)
                chain.expect(centers)
        if centers is None:
            centers = F.furthest_point_sample(coords, self.num_centers)
        # (the max over the neighbours: one streaming pass each way on the GPU, csrc/pool.hip; elsewhere torch.max itself)
        pooled = [neighbor_max(mlp(grouper(coords, centers, features)))
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return (torch.cat(pooled, dim=1) if len(pooled) > 1 else pooled[0]), centers

    def extra_repr(self):
        return f'num_centers={self.num_centers}, out_channels={self.out_channels}'


class PointNetFPModule(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.mlp = SharedMLP(in_channels=in_channels, out_channels=out_channels, dim=1)

    def forward(self, inputs):
        # (points_coords, centers_coords, centers_features[, points_features])
        points_coords, centers_coords, centers_features = inputs[:3]
        skip = inputs[3] if len(inputs) > 3 else None
        lifted = F.nearest_neighbor_interpolate(points_coords, centers_coords, centers_features)
        if skip is not None:
            lifted = torch.cat([lifted, skip], dim=1)
        return self.mlp(lifted), points_coords
