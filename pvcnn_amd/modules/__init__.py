"""Host-side mirror of the reference's `modules` package (modules/__init__.py:1-8): the same
ten public classes with the same constructor / forward signatures and sub-module names
(state_dict keys), implemented over the gfx950 native backend."""
from .ball_query import BallQuery
from .frustum import FrustumPointNetLoss
from .loss import KLLoss
from .pointnet import PointNetAModule, PointNetSAModule, PointNetFPModule
from .pvconv import PVConv
from .se import SE3d
from .shared_mlp import SharedMLP
from .voxelization import Voxelization

__all__ = ['BallQuery', 'FrustumPointNetLoss', 'KLLoss', 'PointNetAModule', 'PointNetSAModule',
           'PointNetFPModule', 'PVConv', 'SE3d', 'SharedMLP', 'Voxelization']
