"""nearest_neighbor_interpolate: 3-NN inverse-squared-distance feature interpolation
(reference: modules/functional/interpolatation.py:8-38; the file name's spelling is the reference's)."""
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['nearest_neighbor_interpolate']


class NeighborInterpolation(Function):
    """points_coords (B,3,N), centers_coords (B,3,M), centers_features (B,C,M) -> (B,C,N).
    Only centers_features receives a gradient."""

    @staticmethod
    @amp_fwd
    def forward(ctx, points_coords, centers_coords, centers_features):
        pts = points_coords.contiguous()
        ctr = centers_coords.contiguous()
        feats = centers_features.contiguous()
        out, nn_idx, nn_w = native().three_nearest_neighbors_interpolate_forward(pts, ctr, feats)
        ctx.save_for_backward(nn_idx, nn_w)
        ctx.num_centers = ctr.size(-1)
        return out

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_points):
        nn_idx, nn_w = ctx.saved_tensors
        grad_centers = native().three_nearest_neighbors_interpolate_backward(
            grad_points.contiguous(), nn_idx, nn_w, ctx.num_centers)
        return None, None, grad_centers


nearest_neighbor_interpolate = NeighborInterpolation.apply
