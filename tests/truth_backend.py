"""fp64 "truth" backend -- TEST INFRASTRUCTURE (used by test_gpu_train_parity.py only).

The 12 native entry points with every floating-point accumulation in float64, on CPU tensors, written with plain
torch indexing.  Everything DISCRETE (voxel ids, trilinear corner indices and their fp32 weights, FPS / ball-query /
3-NN indices and the fp32 3-NN weights) comes from the CPU oracle evaluated in fp32 exactly as the reference kernels
define it, so the truth stack builds the same neighbourhoods as both fp32 stacks; only sums and products are exact.
It answers "who is closer to the exact result": when the HIP path and the fp32 oracle stack disagree end to end,
both are compared with this evaluation of the same network in double precision.
"""
import torch


class TruthBackend:
    name = 'truth-fp64'

    def __init__(self, oracle):
        self.o = oracle

    # ---- voxelize -----------------------------------------------------------------------------------------------
    def avg_voxelize_forward(self, features, coords, resolution):
        b, c, n = features.shape
        r = int(resolution)
        s = r * r * r
        co = coords.long()
        ind = (co[:, 0] * r * r + co[:, 1] * r + co[:, 2]).clamp(0, s - 1)
        cnt = torch.zeros(b, s, dtype=torch.int64).scatter_add_(1, ind, torch.ones_like(ind))
        rcp = (1.0 / cnt.gather(1, ind).float()).to(features.dtype)           # 1/cnt rounded to fp32, like vox.cu:66
        out = torch.zeros(b, c, s, dtype=features.dtype).scatter_add_(2, ind.unsqueeze(1).expand(-1, c, -1), features * rcp.unsqueeze(1))
        return [out, ind.int(), cnt.int()]

    def avg_voxelize_backward(self, grad_y, indices, cnt):
        c = grad_y.shape[1]
        ind = indices.long()
        rcp = (1.0 / cnt.long().gather(1, ind).float()).to(grad_y.dtype)
        return grad_y.gather(2, ind.unsqueeze(1).expand(-1, c, -1)) * rcp.unsqueeze(1)

    # ---- devoxelize ---------------------------------------------------------------------------------------------
    def trilinear_devoxelize_forward(self, r, is_training, coords, features):
        b, c, s = features.shape
        n = coords.shape[2]
        _, inds, wgts = self.o.trilinear_devoxelize_forward(int(r), True, coords.float().contiguous(), torch.zeros(b, 1, s))
        out = torch.zeros(b, c, n, dtype=features.dtype)
        for k in range(8):
            out += features.gather(2, inds[:, k].long().unsqueeze(1).expand(-1, c, -1)) * wgts[:, k].to(features.dtype).unsqueeze(1)
        if is_training:
            return [out, inds, wgts]
        return [out, torch.zeros(1, dtype=torch.int32), torch.zeros(1)]

    def trilinear_devoxelize_backward(self, grad_y, indices, weights, r):
        b, c, n = grad_y.shape
        gx = torch.zeros(b, c, int(r) ** 3, dtype=grad_y.dtype)
        for k in range(8):
            gx.scatter_add_(2, indices[:, k].long().unsqueeze(1).expand(-1, c, -1), grad_y * weights[:, k].to(grad_y.dtype).unsqueeze(1))
        return gx

    # ---- sampling / grouping ------------------------------------------------------------------------------------
    def gather_features_forward(self, features, indices):
        return features.gather(2, indices.long().unsqueeze(1).expand(-1, features.shape[1], -1))

    def gather_features_backward(self, grad_y, indices, n):
        b, c, m = grad_y.shape
        return torch.zeros(b, c, int(n), dtype=grad_y.dtype).scatter_add_(2, indices.long().unsqueeze(1).expand(-1, c, -1), grad_y)

    def furthest_point_sampling(self, coords, num_samples):
        return self.o.furthest_point_sampling(coords.float().contiguous(), num_samples)

    def ball_query(self, centers_coords, points_coords, radius, num_neighbors):
        return self.o.ball_query(centers_coords.float().contiguous(), points_coords.float().contiguous(), radius, num_neighbors)

    def grouping_forward(self, features, indices):
        b, c, n = features.shape
        _, m, u = indices.shape
        flat = indices.long().view(b, 1, m * u).expand(-1, c, -1)
        return features.gather(2, flat).view(b, c, m, u)

    def grouping_backward(self, grad_y, indices, n):
        b, c, m, u = grad_y.shape
        flat = indices.long().view(b, 1, m * u).expand(-1, c, -1)
        return torch.zeros(b, c, int(n), dtype=grad_y.dtype).scatter_add_(2, flat, grad_y.reshape(b, c, m * u))

    def three_nearest_neighbors_interpolate_forward(self, points_coords, centers_coords, centers_features):
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        _, idx, w = self.o.three_nearest_neighbors_interpolate_forward(points_coords.float().contiguous(), centers_coords.float().contiguous(),
                                                                       torch.zeros(b, 1, m))
        out = torch.zeros(b, c, n, dtype=centers_features.dtype)
        for k in range(3):
            out += centers_features.gather(2, idx[:, k].long().unsqueeze(1).expand(-1, c, -1)) * w[:, k].to(out.dtype).unsqueeze(1)
        return [out, idx, w]

    def three_nearest_neighbors_interpolate_backward(self, grad_y, indices, weights, m):
        b, c, n = grad_y.shape
        gx = torch.zeros(b, c, int(m), dtype=grad_y.dtype)
        for k in range(3):
            gx.scatter_add_(2, indices[:, k].long().unsqueeze(1).expand(-1, c, -1), grad_y * weights[:, k].to(grad_y.dtype).unsqueeze(1))
        return gx
