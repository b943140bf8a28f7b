"""BallQuery: neighbourhood grouping around sampled centres (reference: modules/ball_query.py:9-34).

forward(points_coords (B,3,N), centers_coords (B,3,M), points_features (B,C,N) | None)
    -> (B, 3 [+ C], M, U): neighbour coordinates relative to their centre, optionally
       concatenated (coordinates first) with the neighbours' features.
Note the native op takes (centers, points) while this module takes (points, centers)."""
import torch
import torch.nn as nn

from . import functional as F

__all__ = ['BallQuery']


class BallQuery(nn.Module):
    def __init__(self, radius, num_neighbors, include_coordinates=True):
        super().__init__()
        self.radius = radius
        self.num_neighbors = num_neighbors
        self.include_coordinates = include_coordinates

    def forward(self, points_coords, centers_coords, points_features=None):
        points_coords = points_coords.contiguous()
        centers_coords = centers_coords.contiguous()
        nbr = F.ball_query(centers_coords, points_coords, self.radius, self.num_neighbors)
        local_xyz = F.grouping(points_coords, nbr) - centers_coords.unsqueeze(-1)
        if points_features is None:
            assert self.include_coordinates, 'No Features For Grouping'
            return local_xyz
        grouped = F.grouping(points_features, nbr)
        return torch.cat([local_xyz, grouped], dim=1) if self.include_coordinates else grouped

    def extra_repr(self):
        tail = ', include coordinates' if self.include_coordinates else ''
        return f'radius={self.radius}, num_neighbors={self.num_neighbors}{tail}'
