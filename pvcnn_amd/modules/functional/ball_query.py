"""ball_query: first-U-within-radius neighbour indices (reference: modules/functional/ball_query.py:8-19).
Not differentiable; int32 output.  Mind the argument order: (centers, points, radius, U)."""
from ._autograd import native

__all__ = ['ball_query']


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """centers (B,3,M), points (B,3,N) -> IntTensor (B,M,U)."""
    return native().ball_query(centers_coords.float().contiguous(), points_coords.float().contiguous(),
                               radius, num_neighbors)
