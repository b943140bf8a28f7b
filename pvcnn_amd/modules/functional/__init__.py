"""`modules.functional` of the reference (modules/functional/__init__.py:1-7): same ten names."""
from .ball_query import ball_query
from .devoxelization import trilinear_devoxelize
from .grouping import grouping
from .interpolatation import nearest_neighbor_interpolate
from .loss import kl_loss, huber_loss
from .sampling import gather, furthest_point_sample, logits_mask
from .voxelization import avg_voxelize

__all__ = ['ball_query', 'trilinear_devoxelize', 'grouping', 'nearest_neighbor_interpolate', 'kl_loss',
           'huber_loss', 'gather', 'furthest_point_sample', 'logits_mask', 'avg_voxelize']
