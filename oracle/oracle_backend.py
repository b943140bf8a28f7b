"""CPU oracle backend: the reference's 12 native entry points (bindings.cpp:10-37) on CPU tensors.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never from pvcnn_amd/.  It wraps oracle/libpvcnn_oracle.so (the C restatement
in pvcnn_oracle.c) behind the same object shape as the reference's `_backend`
(modules/functional/backend.py:6-25): same function names, argument order, return shapes and
dtypes, same RuntimeError on wrong dtype / non-contiguous input (utils.hpp:7-18; the CHECK_CUDA
check becomes "must be a CPU tensor").  Allocation mirrors the reference host code (torch::zeros
in each *.cpp), so every output is zero-initialised before the C loop runs.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libpvcnn_oracle.so')

_F = ctypes.c_void_p
_I = ctypes.c_int


def build(force=False):
    """Compile the C restatement with gcc (no GPU, no torch involved)."""
    src = os.path.join(_HERE, 'pvcnn_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libpvcnn_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for name in dir(_Sig):
            if name.startswith('orc_'):
                getattr(_lib, name).argtypes = getattr(_Sig, name)
                getattr(_lib, name).restype = None
        _lib.orc_version.restype = _I
        _lib.orc_version.argtypes = []
    return _lib


class _Sig:
    orc_avg_voxelize_fwd = [_F, _F, _I, _I, _I, _I, _F, _F, _F]
    orc_avg_voxelize_fwd_f64 = [_F, _F, _F, _I, _I, _I, _I, _F]
    orc_avg_voxelize_bwd = [_F, _F, _F, _I, _I, _I, _I, _F]
    orc_trilinear_devox_fwd = [_F, _F, _I, _I, _I, _I, _I, _F, _F, _F]
    orc_trilinear_devox_fwd_f64 = [_F, _F, _I, _I, _I, _I, _F]
    orc_trilinear_devox_bwd = [_F, _F, _F, _I, _I, _I, _I, _F]
    orc_trilinear_devox_bwd_f64 = [_F, _F, _F, _I, _I, _I, _I, _F]
    orc_ball_query = [_F, _F, _I, _I, _I, ctypes.c_float, _I, _F]
    orc_grouping_fwd = [_F, _F, _I, _I, _I, _I, _I, _F]
    orc_grouping_bwd = [_F, _F, _I, _I, _I, _I, _I, _F]
    orc_gather_fwd = [_F, _F, _I, _I, _I, _I, _F]
    orc_gather_bwd = [_F, _F, _I, _I, _I, _I, _F]
    orc_fps = [_F, _I, _I, _I, _F, _F]
    orc_three_nn_interp_fwd = [_F, _F, _F, _I, _I, _I, _I, _F, _F, _F]
    orc_three_nn_interp_bwd = [_F, _F, _F, _I, _I, _I, _I, _F]


def _chk(t, name, dtype):
    if t.device.type != 'cpu':
        raise RuntimeError(f'{name} must be a CPU tensor (oracle backend)')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be a contiguous tensor')
    if t.dtype != dtype:
        raise RuntimeError(f'{name} must be a{"n int" if dtype == torch.int32 else " float"} tensor')


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class OracleBackend:
    """Same 12 callables as the reference's pybind module `_pvcnn_backend`."""

    name = 'oracle-cpu'

    # -- sampling.cpp:6-41 ------------------------------------------------------------------
    def gather_features_forward(self, features, indices):
        _chk(features, 'features', torch.float32)
        _chk(indices, 'indices', torch.int32)
        b, c, n = features.shape
        m = indices.shape[1]
        out = torch.zeros(b, c, m, dtype=torch.float32)
        lib().orc_gather_fwd(_p(features), _p(indices), b, c, n, m, _p(out))
        return out

    def gather_features_backward(self, grad_y, indices, n):
        _chk(grad_y, 'grad_y', torch.float32)
        _chk(indices, 'indices', torch.int32)
        b, c, m = grad_y.shape
        grad_x = torch.zeros(b, c, n, dtype=torch.float32)
        lib().orc_gather_bwd(_p(grad_y), _p(indices), b, c, n, m, _p(grad_x))
        return grad_x

    # -- sampling.cpp:43-58 -----------------------------------------------------------------
    def furthest_point_sampling(self, coords, num_samples):
        _chk(coords, 'coords', torch.float32)
        b, _, n = coords.shape
        indices = torch.zeros(b, num_samples, dtype=torch.int32)
        distances = torch.full((b, n), 1e38, dtype=torch.float32)
        lib().orc_fps(_p(coords), b, n, num_samples, _p(distances), _p(indices))
        return indices

    # -- ball_query.cpp:6-30 ----------------------------------------------------------------
    def ball_query(self, centers_coords, points_coords, radius, num_neighbors):
        _chk(centers_coords, 'centers_coords', torch.float32)
        _chk(points_coords, 'points_coords', torch.float32)
        b, _, m = centers_coords.shape
        n = points_coords.shape[2]
        out = torch.zeros(b, m, num_neighbors, dtype=torch.int32)
        r = ctypes.c_float(radius).value                # `const float radius`
        r2 = ctypes.c_float(r * r).value                # radius * radius in float (ball_query.cpp:24)
        lib().orc_ball_query(_p(centers_coords), _p(points_coords), b, n, m, r2, num_neighbors, _p(out))
        return out

    # -- grouping.cpp:6-44 ------------------------------------------------------------------
    def grouping_forward(self, features, indices):
        _chk(features, 'features', torch.float32)
        _chk(indices, 'indices', torch.int32)
        b, c, n = features.shape
        _, m, u = indices.shape
        out = torch.zeros(b, c, m, u, dtype=torch.float32)
        lib().orc_grouping_fwd(_p(features), _p(indices), b, c, n, m, u, _p(out))
        return out

    def grouping_backward(self, grad_y, indices, n):
        _chk(grad_y, 'grad_y', torch.float32)
        _chk(indices, 'indices', torch.int32)
        b, c = grad_y.shape[:2]
        _, m, u = indices.shape
        grad_x = torch.zeros(b, c, n, dtype=torch.float32)
        lib().orc_grouping_bwd(_p(grad_y), _p(indices), b, c, n, m, u, _p(grad_x))
        return grad_x

    # -- neighbor_interpolate.cpp:6-65 ------------------------------------------------------
    def three_nearest_neighbors_interpolate_forward(self, points_coords, centers_coords, centers_features):
        _chk(points_coords, 'points_coords', torch.float32)
        _chk(centers_coords, 'centers_coords', torch.float32)
        _chk(centers_features, 'centers_features', torch.float32)
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        indices = torch.zeros(b, 3, n, dtype=torch.int32)
        weights = torch.zeros(b, 3, n, dtype=torch.float32)
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().orc_three_nn_interp_fwd(_p(points_coords), _p(centers_coords), _p(centers_features),
                                      b, c, m, n, _p(indices), _p(weights), _p(out))
        return [out, indices, weights]

    def three_nearest_neighbors_interpolate_backward(self, grad_y, indices, weights, m):
        _chk(grad_y, 'grad_y', torch.float32)
        _chk(indices, 'indices', torch.int32)
        _chk(weights, 'weights', torch.float32)
        b, c, n = grad_y.shape
        grad_x = torch.zeros(b, c, m, dtype=torch.float32)
        lib().orc_three_nn_interp_bwd(_p(grad_y), _p(indices), _p(weights), b, c, n, m, _p(grad_x))
        return grad_x

    # -- trilinear_devox.cpp:18-91 (argument order: r, is_training, coords, features) ----------
    def trilinear_devoxelize_forward(self, r, is_training, coords, features):
        _chk(features, 'features', torch.float32)
        _chk(coords, 'coords', torch.float32)
        b, c = features.shape[:2]
        n = coords.shape[2]
        outs = torch.zeros(b, c, n, dtype=torch.float32)
        if is_training:
            inds = torch.zeros(b, 8, n, dtype=torch.int32)
            wgts = torch.zeros(b, 8, n, dtype=torch.float32)
        else:
            inds = torch.zeros(1, dtype=torch.int32)
            wgts = torch.zeros(1, dtype=torch.float32)
        lib().orc_trilinear_devox_fwd(_p(coords), _p(features), b, c, n, int(r), int(bool(is_training)),
                                      _p(inds), _p(wgts), _p(outs))
        return [outs, inds, wgts]

    def trilinear_devoxelize_backward(self, grad_y, indices, weights, r):
        _chk(grad_y, 'grad_y', torch.float32)
        _chk(weights, 'weights', torch.float32)
        _chk(indices, 'indices', torch.int32)
        b, c, n = grad_y.shape
        r3 = r * r * r
        grad_x = torch.zeros(b, c, r3, dtype=torch.float32)
        lib().orc_trilinear_devox_bwd(_p(grad_y), _p(indices), _p(weights), b, c, n, r3, _p(grad_x))
        return grad_x

    # -- vox.cpp:17-76 ----------------------------------------------------------------------
    def avg_voxelize_forward(self, features, coords, resolution):
        _chk(features, 'features', torch.float32)
        _chk(coords, 'coords', torch.int32)
        b, c, n = features.shape
        r = int(resolution)
        s = r * r * r
        ind = torch.zeros(b, n, dtype=torch.int32)
        out = torch.zeros(b, c, s, dtype=torch.float32)
        cnt = torch.zeros(b, s, dtype=torch.int32)
        lib().orc_avg_voxelize_fwd(_p(features), _p(coords), b, c, n, r, _p(out), _p(ind), _p(cnt))
        return [out, ind, cnt]

    def avg_voxelize_backward(self, grad_y, indices, cnt):
        _chk(grad_y, 'grad_y', torch.float32)
        _chk(indices, 'indices', torch.int32)
        _chk(cnt, 'cnt', torch.int32)
        b, c, s = grad_y.shape
        n = indices.shape[1]
        grad_x = torch.zeros(b, c, n, dtype=torch.float32)
        lib().orc_avg_voxelize_bwd(_p(grad_y), _p(indices), _p(cnt), b, c, n, s, _p(grad_x))
        return grad_x

    # -- fp64-accumulated companions (not part of the reference API; "who is closer to truth") --
    def avg_voxelize_forward_f64(self, features, ind, cnt):
        b, c, n = features.shape
        s = cnt.shape[1]
        out = torch.zeros(b, c, s, dtype=torch.float64)
        lib().orc_avg_voxelize_fwd_f64(_p(features), _p(ind), _p(cnt), b, c, n, s, _p(out))
        return out

    def trilinear_devoxelize_forward_f64(self, r, coords, features):
        b, c = features.shape[:2]
        n = coords.shape[2]
        outs = torch.zeros(b, c, n, dtype=torch.float64)
        lib().orc_trilinear_devox_fwd_f64(_p(coords), _p(features), b, c, n, int(r), _p(outs))
        return outs

    def trilinear_devoxelize_backward_f64(self, grad_y, indices, weights, r):
        b, c, n = grad_y.shape
        r3 = r * r * r
        grad_x = torch.zeros(b, c, r3, dtype=torch.float64)
        lib().orc_trilinear_devox_bwd_f64(_p(grad_y), _p(indices), _p(weights), b, c, n, r3, _p(grad_x))
        return grad_x


_backend = OracleBackend()
