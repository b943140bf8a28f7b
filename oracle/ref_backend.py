"""The reference's own kernels on the CPU (oracle/_ref, built by build_ref.py) behind the same
12-function `_backend` shape.  TEST INFRASTRUCTURE: used to pin oracle/pvcnn_oracle.c
(tests/test_oracle_vs_ref.py) and optionally as the "reference" CPU baseline.

Host-side allocation mirrors the reference's *.cpp files (torch::zeros / torch::full(1e38))."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path(variant='fma'):
    return os.path.join(_HERE, '_ref', 'libpvcnn_ref_cpu.so' if variant == 'fma' else f'libpvcnn_ref_cpu_{variant}.so')


def available(variant='fma'):
    return os.path.exists(lib_path(variant))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class RefCpuBackend:
    name = 'reference-kernels-on-cpu'

    def __init__(self, variant='fma'):
        self.lib = ctypes.CDLL(lib_path(variant))
        self.lib.ref_ball_query.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 3
        for name, nint in [('ref_avg_voxelize', 4), ('ref_avg_voxelize_grad', 4), ('ref_trilinear_devoxelize', 5),
                           ('ref_trilinear_devoxelize_grad', 4), ('ref_grouping', 5), ('ref_grouping_grad', 5),
                           ('ref_gather_features', 4), ('ref_gather_features_grad', 4), ('ref_furthest_point_sampling', 3),
                           ('ref_three_nn_interpolate', 4), ('ref_three_nn_interpolate_grad', 4)]:
            fn = getattr(self.lib, name)
            nptr = {'ref_avg_voxelize': 5, 'ref_avg_voxelize_grad': 4, 'ref_trilinear_devoxelize': 5,
                    'ref_trilinear_devoxelize_grad': 4, 'ref_grouping': 3, 'ref_grouping_grad': 3, 'ref_gather_features': 3,
                    'ref_gather_features_grad': 3, 'ref_furthest_point_sampling': 3, 'ref_three_nn_interpolate': 6,
                    'ref_three_nn_interpolate_grad': 4}[name]
            fn.argtypes = [ctypes.c_int] * nint + [ctypes.c_void_p] * nptr
            fn.restype = None
        self.lib.ref_ball_query.restype = None

    def gather_features_forward(self, features, indices):
        b, c, n = features.shape; m = indices.shape[1]
        out = torch.zeros(b, c, m)
        self.lib.ref_gather_features(b, c, n, m, _p(features), _p(indices), _p(out))
        return out

    def gather_features_backward(self, grad_y, indices, n):
        b, c, m = grad_y.shape
        gx = torch.zeros(b, c, n)
        self.lib.ref_gather_features_grad(b, c, n, m, _p(grad_y), _p(indices), _p(gx))
        return gx

    def furthest_point_sampling(self, coords, num_samples):
        b, _, n = coords.shape
        indices = torch.zeros(b, num_samples, dtype=torch.int32)
        distances = torch.full((b, n), 1e38)
        self.lib.ref_furthest_point_sampling(b, n, num_samples, _p(coords), _p(distances), _p(indices))
        return indices

    def ball_query(self, centers_coords, points_coords, radius, num_neighbors):
        b, _, m = centers_coords.shape; n = points_coords.shape[2]
        out = torch.zeros(b, m, num_neighbors, dtype=torch.int32)
        r = ctypes.c_float(radius).value
        self.lib.ref_ball_query(b, n, m, ctypes.c_float(r * r).value, num_neighbors, _p(centers_coords), _p(points_coords), _p(out))
        return out

    def grouping_forward(self, features, indices):
        b, c, n = features.shape; _, m, u = indices.shape
        out = torch.zeros(b, c, m, u)
        self.lib.ref_grouping(b, c, n, m, u, _p(features), _p(indices), _p(out))
        return out

    def grouping_backward(self, grad_y, indices, n):
        b, c = grad_y.shape[:2]; _, m, u = indices.shape
        gx = torch.zeros(b, c, n)
        self.lib.ref_grouping_grad(b, c, n, m, u, _p(grad_y), _p(indices), _p(gx))
        return gx

    def three_nearest_neighbors_interpolate_forward(self, points_coords, centers_coords, centers_features):
        b, c, m = centers_features.shape; n = points_coords.shape[2]
        indices = torch.zeros(b, 3, n, dtype=torch.int32); weights = torch.zeros(b, 3, n); out = torch.zeros(b, c, n)
        self.lib.ref_three_nn_interpolate(b, c, m, n, _p(points_coords), _p(centers_coords), _p(centers_features),
                                          _p(indices), _p(weights), _p(out))
        return [out, indices, weights]

    def three_nearest_neighbors_interpolate_backward(self, grad_y, indices, weights, m):
        b, c, n = grad_y.shape
        gx = torch.zeros(b, c, m)
        self.lib.ref_three_nn_interpolate_grad(b, c, n, m, _p(grad_y), _p(indices), _p(weights), _p(gx))
        return gx

    def trilinear_devoxelize_forward(self, r, is_training, coords, features):
        b, c = features.shape[:2]; n = coords.shape[2]
        outs = torch.zeros(b, c, n)
        if is_training:
            inds = torch.zeros(b, 8, n, dtype=torch.int32); wgts = torch.zeros(b, 8, n)
        else:
            inds = torch.zeros(1, dtype=torch.int32); wgts = torch.zeros(1)
        self.lib.ref_trilinear_devoxelize(b, c, n, int(r), int(bool(is_training)), _p(coords), _p(features), _p(inds), _p(wgts), _p(outs))
        return [outs, inds, wgts]

    def trilinear_devoxelize_backward(self, grad_y, indices, weights, r):
        b, c, n = grad_y.shape
        gx = torch.zeros(b, c, r * r * r)
        self.lib.ref_trilinear_devoxelize_grad(b, c, n, r * r * r, _p(indices), _p(weights), _p(grad_y), _p(gx))
        return gx

    def avg_voxelize_forward(self, features, coords, resolution):
        b, c, n = features.shape; r = int(resolution); s = r ** 3
        ind = torch.zeros(b, n, dtype=torch.int32); out = torch.zeros(b, c, s); cnt = torch.zeros(b, s, dtype=torch.int32)
        self.lib.ref_avg_voxelize(b, c, n, r, _p(coords), _p(features), _p(ind), _p(cnt), _p(out))
        return [out, ind, cnt]

    def avg_voxelize_backward(self, grad_y, indices, cnt):
        b, c, s = grad_y.shape; n = indices.shape[1]
        gx = torch.zeros(b, c, n)
        self.lib.ref_avg_voxelize_grad(b, c, n, s, _p(indices), _p(cnt), _p(grad_y), _p(gx))
        return gx
