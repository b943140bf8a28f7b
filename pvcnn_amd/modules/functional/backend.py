"""The native seam: `_backend` with the reference's 12 entry points, served by libpvcnn_hip.so.

Reference: modules/functional/backend.py:6-25 JIT-builds a pybind module `_pvcnn_backend`
(src/bindings.cpp:10-37) with nvcc; every function rejects non-CUDA tensors (utils.hpp:7).
Here the same object shape is implemented over the C ABI in include/pvcnn_hip.h:

  * same names, argument order and return shapes / dtypes as the pybind functions;
  * same input contract: device tensor, contiguous, float32 / int32 -- violations raise
    RuntimeError like TORCH_CHECK does (plus shape checks the reference lacks);
  * outputs are allocated here with torch.empty (the library writes every element) on the
    inputs' device and work is enqueued on torch's current stream -- no host sync;
  * a failed launch raises (the reference prints and exit(-1)s, cuda_utils.cuh:28-37);
  * there is no CPU path: a CPU tensor or a missing library is an error, never a fallback.
"""
import ctypes
import os

import torch

from ... import _lib
from . import _cache

__all__ = ['_backend']


def _dev(t, name):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA (HIP) tensor -- the PVConv hot path has no CPU implementation')


def _f32(t, name):
    _dev(t, name)
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be a contiguous tensor')
    if t.dtype != torch.float32:
        raise RuntimeError(f'{name} must be a float tensor')


def _f32_rows(t, name):
    """float32 (B, C, S) tensor whose C*S rows of a sample are contiguous; samples may be further apart (a channel
    slice of a wider tensor, e.g. a gradient coming out of torch.cat's backward).  -> batch stride in elements."""
    _dev(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f'{name} must be a float tensor')
    b, c, n = t.shape
    ok = (n == 1 or t.stride(2) == 1) and (c == 1 or t.stride(1) == n) and (b == 1 or t.stride(0) >= c * n)
    if not ok:
        raise RuntimeError(f'{name} must be contiguous within each sample')
    return t.stride(0) if b > 1 else c * n


def batch_strided_ok(t):
    """True when _f32_rows would accept t (callers use it to skip a .contiguous() copy)."""
    if t.dim() != 3 or t.dtype != torch.float32:
        return False
    b, c, n = t.shape
    return (n == 1 or t.stride(2) == 1) and (c == 1 or t.stride(1) == n) and (b == 1 or t.stride(0) >= c * n)


def _i32(t, name):
    _dev(t, name)
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be a contiguous tensor')
    if t.dtype != torch.int32:
        raise RuntimeError(f'{name} must be an int tensor')


def _shape(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(None)


_DUMMIES = {}


def _dummies(dev):
    """The 1-element (inds, wgts) the devoxelize forward returns when it does not emit them (trilinear_devox.cpp:45-53): one zeroed
    pair per device, made once -- a training step asks for them once per PVConv layer behind the first of a resolution (two fill
    launches each)."""
    key = (dev.type, dev.index)
    if key not in _DUMMIES:
        _DUMMIES[key] = (torch.zeros((1,), dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.float32, device=dev))
    return _DUMMIES[key]


class _Launch:
    """Device guard + current stream of the tensor's device for one native call."""

    def __init__(self, ref):
        self.device = ref.device
        self.guard = torch.cuda.device(self.device)

    def __enter__(self):
        self.guard.__enter__()
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __exit__(self, *exc):
        return self.guard.__exit__(*exc)


class HipBackend:
    name = 'hip-gfx950'

    def __init__(self):
        self._lib = None

    @property
    def lib(self):
        if self._lib is None:
            self._lib = _lib.load()
        return self._lib

    @staticmethod
    def _scratch(nbytes, device):
        """Caller-owned scratch for the deterministic scatters (torch's caching allocator makes this cheap)."""
        return torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=device)

    # ---- "last workgroup done" tickets (include/pvcnn_hip.h, ABI v11): zeroed words a reduction's launch leaves zeroed, so that its
    # finalize step is the tail of the launch instead of a ~5 us launch of its own.  MEASURED AND NOT THE DEFAULT (round 5,
    # profiles/ab/r05d_fold_without_fences.md): the published atomics + tickets of ~100 k workgroups per step cost what the 26 launches
    # cost (PVCNN 6.636 vs 6.605 ms, Frustum-PVCNN 6.19 vs 6.07 ms); with fences, twice the step.  PVCNN_FOLD_FINALIZE=1 (read once per
    # process) switches it on; the tests pin both paths against each other.  One persistent pool per device (a captured graph keeps the
    # addresses); slices are handed out round robin PER (device, stream) -- launches on one stream never overlap, launches on two
    # streams (a side stream, a parallel branch of a captured step) draw from pools of their own (a word shared by two
    # concurrent launches would fool the "last workgroup" decision)
    fold_finalize = os.environ.get('PVCNN_FOLD_FINALIZE', '0') == '1'
    _TICKET_POOL = 1 << 16

    def _tickets(self, n, device):
        """-> a zeroed int32 view of n words (every kernel that takes tickets leaves them zeroed), or None when folding is off."""
        if not self.fold_finalize or n > self._TICKET_POOL:
            return None
        pools = self.__dict__.setdefault('_ticket_pools', {})
        key = (device.type, device.index, int(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else 0)
        ent = pools.get(key)
        if ent is None:
            ent = pools[key] = [torch.zeros((self._TICKET_POOL,), dtype=torch.int32, device=device), 0]
        if ent[1] + n > self._TICKET_POOL:
            ent[1] = 0
        view = ent[0][ent[1]:ent[1] + n]
        ent[1] += (int(n) + 15) & ~15
        return view

    # ---- sampling.cpp:6-41 ------------------------------------------------------------------
    def gather_features_forward(self, features, indices):
        _f32(features, 'features'); _i32(indices, 'indices')
        _shape(features.dim() == 3 and indices.dim() == 2 and indices.shape[0] == features.shape[0],
               'gather: features (B,C,N), indices (B,M) expected')
        b, c, n = features.shape
        m = indices.shape[1]
        out = torch.empty((b, c, m), dtype=torch.float32, device=features.device)
        with _Launch(features) as s:
            _lib.check(self.lib.pvcnn_gather_fwd(_p(features), _p(indices), b, c, n, m, _p(out), s), 'gather_features_forward')
        return out

    def gather_features_backward(self, grad_y, indices, n):
        _f32(grad_y, 'grad_y'); _i32(indices, 'indices')
        _shape(grad_y.dim() == 3 and indices.dim() == 2 and indices.shape == (grad_y.shape[0], grad_y.shape[2]),
               'gather backward: grad_y (B,C,M), indices (B,M) expected')
        b, c, m = grad_y.shape
        grad_x = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_y.device)
        ws = self._scratch(self.lib.pvcnn_gather_bwd_workspace_bytes(b, c, int(n), m), grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_gather_bwd(_p(grad_y), _p(indices), b, c, int(n), m, _p(grad_x), _p(ws), ws.numel(), s),
                       'gather_features_backward')
        return grad_x

    # ---- sampling.cpp:43-58 -----------------------------------------------------------------
    def furthest_point_sampling(self, coords, num_samples):
        _f32(coords, 'coords')
        _shape(coords.dim() == 3 and coords.shape[1] == 3, 'furthest_point_sampling: coords (B,3,N) expected')
        b, _, n = coords.shape
        m = int(num_samples)
        indices = torch.empty((b, m), dtype=torch.int32, device=coords.device)
        distances = None
        if n > 16384:   # PVCNN_FPS_MAX_RESIDENT_POINTS: larger clouds need the global scratch
            distances = torch.empty((b, n), dtype=torch.float32, device=coords.device)
        with _Launch(coords) as s:
            _lib.check(self.lib.pvcnn_fps(_p(coords), b, n, m, _p(distances), _p(indices), s), 'furthest_point_sampling')
        return indices

    # ---- the index selection of logits_mask (modules/functional/sampling.py:69-82) on the device ----------------
    has_mask_select = True

    def mask_select(self, mask, num_samples, choices=None, seed=None):
        """mask (B,N) bool -> selected (B,M) int32 foreground point ids (see include/pvcnn_hip.h).  choices (B,M) int32:
        parity mode (the caller's draws); else seed: int64 device tensor (key, stream id) for the Philox stream."""
        _dev(mask, 'mask')
        _shape(mask.dim() == 2 and mask.dtype in (torch.bool, torch.uint8) and mask.is_contiguous(), 'mask_select: mask (B,N) bool expected')
        b, n = mask.shape
        m = int(num_samples)
        if choices is not None:
            _i32(choices, 'choices')
            _shape(tuple(choices.shape) == (b, m), 'mask_select: choices (B,M) expected')
        else:
            _dev(seed, 'seed')
            _shape(seed.dtype == torch.int64 and seed.numel() >= 2 and seed.is_contiguous(), 'mask_select: seed = 2 x int64 on the device')
        selected = torch.empty((b, m), dtype=torch.int32, device=mask.device)
        with _Launch(mask) as s:
            _lib.check(self.lib.pvcnn_mask_select(_p(mask.view(torch.uint8)), b, n, m, _p(choices) if choices is not None else None,
                                                  _p(seed) if choices is None else None, _p(selected), None, s), 'mask_select')
        return selected

    # ---- ball_query.cpp:6-30 ----------------------------------------------------------------
    def ball_query(self, centers_coords, points_coords, radius, num_neighbors):
        _f32(centers_coords, 'centers_coords'); _f32(points_coords, 'points_coords')
        _shape(centers_coords.dim() == 3 and points_coords.dim() == 3 and centers_coords.shape[1] == 3
               and points_coords.shape[1] == 3 and centers_coords.shape[0] == points_coords.shape[0],
               'ball_query: centers (B,3,M), points (B,3,N) expected')
        b, _, m = centers_coords.shape
        n = points_coords.shape[2]
        u = int(num_neighbors)
        out = torch.empty((b, m, u), dtype=torch.int32, device=centers_coords.device)
        with _Launch(centers_coords) as s:
            _lib.check(self.lib.pvcnn_ball_query(_p(centers_coords), _p(points_coords), b, n, m, float(radius), u, _p(out), s), 'ball_query')
        return out

    # ---- grouping.cpp:6-44 ------------------------------------------------------------------
    def grouping_forward(self, features, indices):
        _f32(features, 'features'); _i32(indices, 'indices')
        _shape(features.dim() == 3 and indices.dim() == 3 and indices.shape[0] == features.shape[0],
               'grouping: features (B,C,N), indices (B,M,U) expected')
        b, c, n = features.shape
        _, m, u = indices.shape
        out = torch.empty((b, c, m, u), dtype=torch.float32, device=features.device)
        with _Launch(features) as s:
            _lib.check(self.lib.pvcnn_grouping_fwd(_p(features), _p(indices), b, c, n, m, u, _p(out), s), 'grouping_forward')
        return out

    def grouping_backward(self, grad_y, indices, n):
        _f32(grad_y, 'grad_y'); _i32(indices, 'indices')
        _shape(grad_y.dim() == 4 and indices.dim() == 3 and tuple(indices.shape) == (grad_y.shape[0], grad_y.shape[2], grad_y.shape[3]),
               'grouping backward: grad_y (B,C,M,U), indices (B,M,U) expected')
        b, c, m, u = grad_y.shape
        grad_x = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_y.device)
        ws = self._scratch(self.lib.pvcnn_grouping_bwd_workspace_bytes(b, c, int(n), m, u), grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_grouping_bwd(_p(grad_y), _p(indices), b, c, int(n), m, u, _p(grad_x), _p(ws), ws.numel(), s),
                       'grouping_backward')
        return grad_x

    # ---- neighbor_interpolate.cpp:6-65 ------------------------------------------------------
    def three_nearest_neighbors_interpolate_forward(self, points_coords, centers_coords, centers_features):
        _f32(points_coords, 'points_coords'); _f32(centers_coords, 'centers_coords'); _f32(centers_features, 'centers_features')
        _shape(points_coords.dim() == 3 and centers_coords.dim() == 3 and centers_features.dim() == 3
               and points_coords.shape[1] == 3 and centers_coords.shape[1] == 3
               and centers_coords.shape[2] == centers_features.shape[2]
               and points_coords.shape[0] == centers_coords.shape[0] == centers_features.shape[0],
               '3-NN interpolate: points (B,3,N), centers (B,3,M), features (B,C,M) expected')
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        dev = points_coords.device
        indices = torch.empty((b, 3, n), dtype=torch.int32, device=dev)
        weights = torch.empty((b, 3, n), dtype=torch.float32, device=dev)
        out = torch.empty((b, c, n), dtype=torch.float32, device=dev)
        with _Launch(points_coords) as s:
            _lib.check(self.lib.pvcnn_three_nn_interp_fwd(_p(points_coords), _p(centers_coords), _p(centers_features),
                                                          b, c, m, n, _p(indices), _p(weights), _p(out), s),
                       'three_nearest_neighbors_interpolate_forward')
        return [out, indices, weights]

    def three_nearest_neighbors_interpolate_backward(self, grad_y, indices, weights, m):
        _f32(grad_y, 'grad_y'); _i32(indices, 'indices'); _f32(weights, 'weights')
        _shape(grad_y.dim() == 3 and tuple(indices.shape) == (grad_y.shape[0], 3, grad_y.shape[2])
               and indices.shape == weights.shape,
               '3-NN interpolate backward: grad_y (B,C,N), indices/weights (B,3,N) expected')
        b, c, n = grad_y.shape
        grad_x = torch.empty((b, c, int(m)), dtype=torch.float32, device=grad_y.device)
        ws = self._scratch(self.lib.pvcnn_three_nn_interp_bwd_workspace_bytes(b, c, n, int(m)), grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_three_nn_interp_bwd(_p(grad_y), _p(indices), _p(weights), b, c, n, int(m), _p(grad_x),
                                                          _p(ws), ws.numel(), s),
                       'three_nearest_neighbors_interpolate_backward')
        return grad_x

    # the two scatter entries of the 12-callable seam keep their plan on the tensor they were called with (False: the one-shot C entries,
    # which rebuild the counting sort per call -- kept reachable for the tests that pin them)
    seam_plan_memo = True

    # ---- trilinear_devox.cpp:18-91 (argument order: r, is_training, coords, features) ----------
    def trilinear_devoxelize_forward(self, r, is_training, coords, features):
        _f32(features, 'features'); _f32(coords, 'coords')
        r = int(r)
        _shape(features.dim() == 3 and coords.dim() == 3 and coords.shape[1] == 3
               and coords.shape[0] == features.shape[0] and features.shape[2] == r * r * r,
               'trilinear_devoxelize: coords (B,3,N), features (B,C,R^3) expected')
        b, c = features.shape[:2]
        n = coords.shape[2]
        dev = features.device
        outs = torch.empty((b, c, n), dtype=torch.float32, device=dev)
        if is_training:
            inds = torch.empty((b, 8, n), dtype=torch.int32, device=dev)
            wgts = torch.empty((b, 8, n), dtype=torch.float32, device=dev)
        else:   # 1-element dummies, like trilinear_devox.cpp:45-53
            inds, wgts = _dummies(dev)
        with _Launch(features) as s:
            _lib.check(self.lib.pvcnn_trilinear_devox_fwd(_p(coords), _p(features), b, c, n, r, int(bool(is_training)),
                                                          _p(inds) if is_training else None,
                                                          _p(wgts) if is_training else None, _p(outs), s),
                       'trilinear_devoxelize_forward')
        return [outs, inds, wgts]

    def trilinear_devoxelize_backward(self, grad_y, indices, weights, r):
        gy_bstride = _f32_rows(grad_y, 'grad_y'); _f32(weights, 'weights'); _i32(indices, 'indices')
        _shape(grad_y.dim() == 3 and tuple(indices.shape) == (grad_y.shape[0], 8, grad_y.shape[2])
               and indices.shape == weights.shape,
               'trilinear_devoxelize backward: grad_y (B,C,N), indices/weights (B,8,N) expected')
        b, c, n = grad_y.shape
        r = int(r)
        # PLAN REUSE BEHIND THE REFERENCE'S OWN CALL (functional/devoxelization.py:30-39 passes the saved (inds, wgts) here on every
        # backward): the counting sort depends on (inds, wgts, R) only, so it is memoised on the `indices` tensor OBJECT (identity +
        # in-place version + storage view: _cache.memo; dies with the tensor) together with the identity / version of `weights` -- a
        # second call with the same saved tensors runs the apply alone.  No new API: section B users of INTEGRATION.md get it as is.
        if self.seam_plan_memo and self.lib.pvcnn_trilinear_devox_bwd_plan_bytes(b, n, r):
            key = ('seam devox bwd plan', r, id(weights), weights._version, weights.data_ptr())
            plan = _cache.memo(indices, key, lambda: self.trilinear_devoxelize_backward_plan(indices, weights, r))
            if plan is not None:
                return self.trilinear_devoxelize_backward_apply(grad_y, plan, r)
        grad_x = torch.empty((b, c, r * r * r), dtype=torch.float32, device=grad_y.device)
        ws = self._scratch(self.lib.pvcnn_trilinear_devox_bwd_workspace_bytes(b, c, n, r), grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_trilinear_devox_bwd_strided(_p(grad_y), gy_bstride, _p(indices), _p(weights), b, c, n, r,
                                                                  _p(grad_x), _p(ws), ws.numel(), s),
                       'trilinear_devoxelize_backward')
        return grad_x

    # ---- modules/voxelization.py:16-25 ---------------------------------------------------------
    has_voxel_coords = True

    def voxel_coords(self, coords, resolution, normalize, eps):
        """coords (B,3,N) -> (norm_coords float (B,3,N) in [0,R-1], vox_coords int32 (B,3,N)) in one launch."""
        _f32(coords, 'coords')
        _shape(coords.dim() == 3 and coords.shape[1] == 3, 'voxel_coords: coords (B,3,N) expected')
        b, _, n = coords.shape
        norm = torch.empty_like(coords)
        vox = torch.empty((b, 3, n), dtype=torch.int32, device=coords.device)
        with _Launch(coords) as s:
            _lib.check(self.lib.pvcnn_voxel_coords(_p(coords), b, n, int(resolution), int(bool(normalize)), float(eps),
                                                   _p(norm), _p(vox), s), 'voxel_coords')
        return norm, vox

    def voxel_coords_tail(self, coords, mean, radius, resolution, eps):
        """The fused elementwise tail of Voxelization.forward: coords (B,3,N) (rows contiguous within a cloud; may be a
        channel slice of a wider tensor), mean (B,3,1) and radius (B,1,1) | None from the reference's own torch
        reductions -> (norm_coords, vox_coords), bit-identical to modules/voxelization.py:16-25 on this device."""
        cstride = _f32_rows(coords, 'coords')
        _shape(coords.dim() == 3 and coords.shape[1] == 3, 'voxel_coords: coords (B,3,N) expected')
        b, _, n = coords.shape
        _f32(mean, 'mean')
        _shape(mean.numel() == b * 3, 'voxel_coords: mean (B,3) expected')
        if radius is not None:
            _f32(radius, 'radius')
            _shape(radius.numel() == b, 'voxel_coords: radius (B) expected')
        norm = torch.empty((b, 3, n), dtype=torch.float32, device=coords.device)
        vox = torch.empty((b, 3, n), dtype=torch.int32, device=coords.device)
        with _Launch(coords) as s:
            _lib.check(self.lib.pvcnn_voxel_coords_tail(_p(coords), cstride, _p(mean), _p(radius) if radius is not None else None,
                                                        b, n, int(resolution), float(eps), _p(norm), _p(vox), s), 'voxel_coords_tail')
        return norm, vox

    # ---- vox.cpp:17-76 ----------------------------------------------------------------------
    def avg_voxelize_forward(self, features, coords, resolution):
        _f32(features, 'features'); _i32(coords, 'coords')
        _shape(features.dim() == 3 and coords.dim() == 3 and coords.shape[1] == 3
               and coords.shape[0] == features.shape[0] and coords.shape[2] == features.shape[2],
               'avg_voxelize: features (B,C,N), coords (B,3,N) expected')
        b, c, n = features.shape
        r = int(resolution)
        s3 = r * r * r
        dev = features.device
        # plan reuse behind the reference's own call (functional/voxelization.py:10-24): memoised on the int32 `coords` tensor object
        # (identity + in-place version: a coords tensor written in place misses); the plan's ind / cnt are handed out as they are
        # (read-only by the reference's convention: saved for backward) and re-checked by their own version counters on every hit
        if self.seam_plan_memo and self.lib.pvcnn_avg_voxelize_plan_bytes(b, n, r):
            key = ('seam voxelize plan', r)
            hit = _cache.memo(coords, key, lambda: self._stamped_voxel_plan(coords, r))
            if hit is not None and (hit[0].ind._version, hit[0].cnt._version) != hit[1]:     # a caller wrote into ind / cnt: rebuild
                _cache.forget(coords, key)
                hit = _cache.memo(coords, key, lambda: self._stamped_voxel_plan(coords, r))
            if hit is not None:
                return [self.avg_voxelize_apply(features, hit[0]), hit[0].ind, hit[0].cnt]
        out = torch.empty((b, c, s3), dtype=torch.float32, device=dev)
        ind = torch.empty((b, n), dtype=torch.int32, device=dev)
        cnt = torch.empty((b, s3), dtype=torch.int32, device=dev)
        ws = self._scratch(self.lib.pvcnn_avg_voxelize_fwd_workspace_bytes(b, c, n, r), dev)
        with _Launch(features) as s:
            _lib.check(self.lib.pvcnn_avg_voxelize_fwd(_p(features), _p(coords), b, c, n, r, _p(out), _p(ind), _p(cnt),
                                                       _p(ws), ws.numel(), s), 'avg_voxelize_forward')
        return [out, ind, cnt]

    def _stamped_voxel_plan(self, coords, r):
        vp = self.avg_voxelize_plan(coords, r)
        return None if vp is None else (vp, (vp.ind._version, vp.cnt._version))

    def avg_voxelize_backward(self, grad_y, indices, cnt):
        _f32(grad_y, 'grad_y'); _i32(indices, 'indices'); _i32(cnt, 'cnt')
        _shape(grad_y.dim() == 3 and indices.dim() == 2 and tuple(cnt.shape) == (grad_y.shape[0], grad_y.shape[2])
               and indices.shape[0] == grad_y.shape[0],
               'avg_voxelize backward: grad_y (B,C,S), indices (B,N), cnt (B,S) expected')
        b, c, s3 = grad_y.shape
        n = indices.shape[1]
        grad_x = torch.empty((b, c, n), dtype=torch.float32, device=grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_avg_voxelize_bwd(_p(grad_y), _p(indices), _p(cnt), b, c, n, s3, _p(grad_x), s),
                       'avg_voxelize_backward')
        return grad_x


    # ---- scatter plans (include/pvcnn_hip.h "scatter plans"): one counting sort per (coords, R), applied by every layer ----
    has_scatter_plans = True

    class VoxelPlan:
        """avg_voxelize's plan for one (voxel coordinates, R): ind (B,N), cnt (B,R^3) and the opaque sort plan."""
        __slots__ = ('ind', 'cnt', 'plan', 'r', 'n', 'b')

    def avg_voxelize_plan(self, coords, resolution):
        """coords (B,3,N) int32 -> VoxelPlan, or None when the grid is too large for a plan (one-shot path then)."""
        _i32(coords, 'coords')
        _shape(coords.dim() == 3 and coords.shape[1] == 3, 'avg_voxelize: coords (B,3,N) expected')
        b, _, n = coords.shape
        r = int(resolution)
        nbytes = self.lib.pvcnn_avg_voxelize_plan_bytes(b, n, r)
        if nbytes == 0:
            return None
        dev = coords.device
        vp = self.VoxelPlan()
        vp.r, vp.n, vp.b = r, n, b
        vp.ind = torch.empty((b, n), dtype=torch.int32, device=dev)
        vp.cnt = torch.empty((b, r * r * r), dtype=torch.int32, device=dev)
        vp.plan = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        scratch = self._scratch(self.lib.pvcnn_avg_voxelize_plan_scratch_bytes(b, n, r), dev)
        with _Launch(coords) as s:
            _lib.check(self.lib.pvcnn_avg_voxelize_plan(_p(coords), b, n, r, _p(vp.ind), _p(vp.cnt), _p(vp.plan), vp.plan.numel(),
                                                        _p(scratch), scratch.numel(), s), 'avg_voxelize_plan')
        return vp

    # PVCNN_PAIR_PLANS=0 (read once per process): the two plans of a PVConv geometry from their own chains (A/B; the pair is the default)
    has_pvconv_plans = os.environ.get('PVCNN_PAIR_PLANS', '1') != '0'

    def pvconv_plans(self, vox_coords, norm_coords, resolution):
        """(vox_coords int32 (B,3,N), norm_coords float (B,3,N), R) -> (VoxelPlan, devoxelize-backward plan tensor): what
        avg_voxelize_plan(vox_coords, R) and trilinear_devoxelize_backward_plan(inds, wgts, R) of the taps at norm_coords build, from ONE
        chain of three launches (csrc/csr.h: launch_csr_prep_pair).  None when the grid is too large for a plan."""
        _i32(vox_coords, 'vox_coords'); _f32(norm_coords, 'norm_coords')
        _shape(vox_coords.dim() == 3 and vox_coords.shape[1] == 3 and tuple(norm_coords.shape) == tuple(vox_coords.shape),
               'pvconv_plans: vox_coords / norm_coords (B,3,N) expected')
        b, _, n = vox_coords.shape
        r = int(resolution)
        vbytes, dbytes = self.lib.pvcnn_avg_voxelize_plan_bytes(b, n, r), self.lib.pvcnn_trilinear_devox_bwd_plan_bytes(b, n, r)
        if vbytes == 0 or dbytes == 0 or n == 0:
            return None
        dev = vox_coords.device
        vp = self.VoxelPlan()
        vp.r, vp.n, vp.b = r, n, b
        vp.ind = torch.empty((b, n), dtype=torch.int32, device=dev)
        vp.cnt = torch.empty((b, r * r * r), dtype=torch.int32, device=dev)
        vp.plan = torch.empty((vbytes,), dtype=torch.uint8, device=dev)
        dplan = torch.empty((dbytes,), dtype=torch.uint8, device=dev)
        scratch = self._scratch(self.lib.pvcnn_pvconv_plans_scratch_bytes(b, n, r), dev)
        with _Launch(vox_coords) as s:
            _lib.check(self.lib.pvcnn_pvconv_plans(_p(vox_coords), _p(norm_coords), b, n, r, _p(vp.ind), _p(vp.cnt), _p(vp.plan), vp.plan.numel(),
                                                   _p(dplan), dplan.numel(), _p(scratch), scratch.numel(), s), 'pvconv_plans')
        return vp, dplan

    def avg_voxelize_apply(self, features, vp):
        """features (B,C,N) -> out (B,C,R^3) with the plan of avg_voxelize_plan (same B, N, R)."""
        _f32(features, 'features')
        _shape(features.dim() == 3 and features.shape[0] == vp.b and features.shape[2] == vp.n, 'avg_voxelize: features do not match the plan')
        b, c, n = features.shape
        out = torch.empty((b, c, vp.r ** 3), dtype=torch.float32, device=features.device)
        with _Launch(features) as s:
            _lib.check(self.lib.pvcnn_avg_voxelize_apply(_p(features), _p(vp.plan), vp.plan.numel(), b, c, n, vp.r, _p(out), s),
                       'avg_voxelize_apply')
        return out

    def trilinear_devoxelize_backward_plan(self, indices, weights, r):
        """(inds, wgts) (B,8,N) -> opaque plan tensor of the backward scatter, or None when R is too large for one."""
        _i32(indices, 'indices'); _f32(weights, 'weights')
        _shape(indices.dim() == 3 and indices.shape[1] == 8 and indices.shape == weights.shape, 'inds / wgts (B,8,N) expected')
        b, _, n = indices.shape
        r = int(r)
        nbytes = self.lib.pvcnn_trilinear_devox_bwd_plan_bytes(b, n, r)
        if nbytes == 0:
            return None
        plan = torch.empty((nbytes,), dtype=torch.uint8, device=indices.device)
        scratch = self._scratch(self.lib.pvcnn_trilinear_devox_bwd_plan_scratch_bytes(b, n, r), indices.device)
        with _Launch(indices) as s:
            _lib.check(self.lib.pvcnn_trilinear_devox_bwd_plan(_p(indices), _p(weights), b, n, r, _p(plan), plan.numel(),
                                                               _p(scratch), scratch.numel(), s), 'trilinear_devoxelize_backward_plan')
        return plan

    def trilinear_devoxelize_backward_apply(self, grad_y, plan, r):
        gy_bstride = _f32_rows(grad_y, 'grad_y')
        b, c, n = grad_y.shape
        r = int(r)
        grad_x = torch.empty((b, c, r * r * r), dtype=torch.float32, device=grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_trilinear_devox_bwd_apply(_p(grad_y), gy_bstride, _p(plan), plan.numel(), b, c, n, r,
                                                                _p(grad_x), s), 'trilinear_devoxelize_backward_apply')
        return grad_x

    # ---- voxel_layers Conv3d (k=3, stride 1, pad 1): modules/pvconv.py:20-27 (cuDNN in the reference) ----
    has_conv3d = True

    def _conv_wt(self, weight, for_bwd_data):
        co, ci = weight.shape[0], weight.shape[1]
        wt = torch.empty((ci * 27 * co,), dtype=torch.float32, device=weight.device)
        with _Launch(weight) as s:
            _lib.check(self.lib.pvcnn_conv3d_weight_transform(_p(weight), co, ci, int(for_bwd_data), _p(wt), s), 'conv3d_weight_transform')
        return wt

    def conv3d_forward(self, x, weight, bias, want_stats=False):
        _f32(x, 'x'); _f32(weight, 'weight')
        _shape(x.dim() == 5 and weight.dim() == 5 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[1] == x.shape[1]
               and x.shape[2] == x.shape[3] == x.shape[4], 'conv3d: x (B,Ci,R,R,R), weight (Co,Ci,3,3,3) expected')
        if bias is not None:
            _f32(bias, 'bias')
        b, ci, r = x.shape[0], x.shape[1], x.shape[2]
        co = weight.shape[0]
        wt = self._conv_wt(weight, False)
        y = torch.empty((b, co, r, r, r), dtype=torch.float32, device=x.device)
        if want_stats:   # per-workgroup (sum, sum of squares) partials for the BatchNorm that follows
            part = torch.empty((co, self.lib.pvcnn_conv3d_fwd_stats_parts(b, co, r), 2), dtype=torch.float32, device=x.device)
            with _Launch(x) as s:
                _lib.check(self.lib.pvcnn_conv3d_fwd_stats(_p(x), _p(wt), _p(bias) if bias is not None else None, b, ci, co, r,
                                                           _p(y), _p(part), s), 'conv3d_forward')
            return y, part
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_conv3d_fwd(_p(x), _p(wt), _p(bias) if bias is not None else None, b, ci, co, r, _p(y), s), 'conv3d_forward')
        return y

    def conv3d_backward_data(self, grad_y, weight):
        _f32(grad_y, 'grad_y'); _f32(weight, 'weight')
        b, co, r = grad_y.shape[0], grad_y.shape[1], grad_y.shape[2]
        ci = weight.shape[1]
        wt = self._conv_wt(weight, True)
        gx = torch.empty((b, ci, r, r, r), dtype=torch.float32, device=grad_y.device)
        with _Launch(grad_y) as s:   # a convolution with Ci and Co exchanged on the flipped weights
            _lib.check(self.lib.pvcnn_conv3d_fwd(_p(grad_y), _p(wt), None, b, co, ci, r, _p(gx), s), 'conv3d_backward_data')
        return gx

    # the backward-weight / BatchNorm-backward entries take `out_w` / `out_b`: where to write the two parameter gradients (contiguous
    # fp32 tensors of the right shape, e.g. the parameter's slot in a flat gradient bucket: functional/_gradslots.py) instead of fresh ones
    has_grad_out = True

    @staticmethod
    def _grad_out(dst, shape, device):
        if dst is None:
            return torch.empty(shape, dtype=torch.float32, device=device)
        _shape(tuple(dst.shape) == tuple(shape) and dst.dtype == torch.float32 and dst.is_contiguous() and dst.device == device,
               'gradient destination: contiguous float32 tensor of the gradient\'s shape expected')
        return dst

    def conv3d_backward_weight(self, x, grad_y, with_bias=False, out_w=None, out_b=None):
        """-> grad_weight, or (grad_weight, grad_bias) when with_bias (the bias sum rides on the same pass)."""
        _f32(x, 'x'); _f32(grad_y, 'grad_y')
        b, ci, r = x.shape[0], x.shape[1], x.shape[2]
        co = grad_y.shape[1]
        gw = self._grad_out(out_w, (co, ci, 3, 3, 3), x.device)
        gb = self._grad_out(out_b, (co,), x.device) if with_bias else None
        ws = self._scratch(self.lib.pvcnn_conv3d_bwd_weight_workspace_bytes(b, ci, co, r), x.device)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_conv3d_bwd_weight(_p(x), _p(grad_y), b, ci, co, r, _p(gw), _p(gb) if with_bias else None,
                                                        _p(ws), ws.numel(), s), 'conv3d_backward_weight')
        return (gw, gb) if with_bias else gw


    # ---- the same convolution on the bf16 matrix cores (csrc/conv3d_bf16.hip): nsplit = 3 "bf16x3" (fp32-class accuracy, up to
    # 2.7x the fp32-MFMA rate) or nsplit = 1 (plain bf16 operands: the autocast / BASELINE configs[4] path) ----
    has_conv3d_split = True
    # default arithmetic of the voxel convolutions' forward / backward-data products:
    #   'f16x2'  : power-of-two scaled fp16 hi + lo split of both fp32 operands, 3 partial products, fp32 accumulate (<= 1e-5 vs fp64)
    #   'bf16x3' : exact three-way bf16 split of both fp32 operands, 6 partial products, fp32 accumulate (<= 1e-5 vs fp64)
    #   'fp32'   : v_mfma_f32_32x32x2_f32, one rounding per product (conv3d.hip)
    conv_math = os.environ.get('PVCNN_CONV_MATH', 'f16x2')
    CONV_NSPLIT = {'f16x2': 2, 'bf16x3': 3, 'fp32': 0}

    def absmax_bits(self, x):
        """One uint32 on the device: the bit pattern of max |x| (a 1-word amax buffer: the scalar scale of the f16x2 kernels)."""
        _f32(x, 'x')
        out = torch.empty((1,), dtype=torch.int32, device=x.device)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_absmax_bits(_p(x), x.numel(), _p(out), s), 'absmax_bits')
        return out

    # "amax buffers" (include/pvcnn_hip.h): [0] = bits of max |x| over the tensor, [1 + t] = bits of the maximum over all channels of
    # position segment t.  The f16x2 forward / backward-data kernels scale each workgroup's tile of x by the maxima of the segments it
    # touches (an outlier costs precision only inside its own tile); the backward-weight kernels scale by the global maximum, which
    # they take from the table, too (ABI v12) -- so a buffer made with want_global=False has NO word [0] and costs no launch for it.
    PW_AMAX_SEG = 256          # the 1x1 GEMM's point tile
    _TABLE_ONLY = ctypes.c_void_p(1)      # PVCNN_TABLE_ONLY (include/pvcnn_hip.h)
    amax_global = os.environ.get('PVCNN_AMAX_GLOBAL', '0') == '1'      # A/B switch: every amax buffer with its word [0] (ABI v11 behaviour: 11 more launches per PVCNN step)

    def absmax_tiles(self, x, seg, want_global=True):
        """x (B, C, L) -> int32 (1 + B * ceil(L / seg),): the amax buffer of x with segments of `seg` positions (one read of x).
        want_global=False: the table only, word [0] is left unwritten (for consumers that are handed the segment length)."""
        _f32(x, 'x')
        _shape(x.dim() == 3, 'absmax_tiles: x (B,C,L) expected')
        b, c, n = x.shape
        seg = int(seg)
        out = torch.empty((self.lib.pvcnn_absmax_tiles_count(b, n, seg),), dtype=torch.int32, device=x.device)
        ticket = self._tickets(1, x.device)
        want_global = want_global or self.amax_global
        tk = (_p(ticket) if ticket is not None else None) if want_global else self._TABLE_ONLY
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_absmax_tiles(_p(x), b, c, n, seg, _p(out), tk, s), 'absmax_tiles')
        return out

    def conv_amax(self, x, want_global=True):
        """amax buffer of a voxel grid x (B,C,R,R,R) in the layout the Conv3d kernels take: one maximum per z row."""
        return self.absmax_tiles(x.view(x.shape[0], x.shape[1], -1), x.shape[2], want_global)

    def pw_amax(self, x, want_global=True):
        """amax buffer of point features x (B,C,N) in the layout the 1x1 GEMM takes: one maximum per 256-point tile."""
        return self.absmax_tiles(x, self.PW_AMAX_SEG, want_global)

    @staticmethod
    def _amax_seg(amax, tiles, seg):
        """0 when `amax` is a 1-word buffer (scalar scale), `seg` when it carries the `tiles`-entry table; anything else is an error."""
        if amax.numel() == 1:
            return 0
        _shape(amax.numel() == 1 + tiles, f'amax buffer has {amax.numel()} words, expected 1 or {1 + tiles}')
        return seg

    def _conv_wsplit(self, weight, for_bwd_data, nsplit):
        co, ci = weight.shape[0], weight.shape[1]
        nbytes = self.lib.pvcnn_conv3d_weight_split_bytes(co, ci, int(for_bwd_data), int(nsplit))
        wts = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
        with _Launch(weight) as s:
            _lib.check(self.lib.pvcnn_conv3d_weight_split(_p(weight), co, ci, int(for_bwd_data), int(nsplit), _p(wts), s), 'conv3d_weight_split')
        return wts

    # ---- the f16x2 weight images of a whole model, refreshed by one launch per kind and step ---------------------------------
    has_weight_bank = True

    def weight_bank_register(self, model):
        """Make the Conv3d / 1x1-convolution weights of `model` eligible for the batched refresh (weight_bank_refresh)."""
        self._bank().register(model)

    def weight_bank_refresh(self):
        """Recompute, in ONE launch per kind, both f16x2 images of every registered weight a forward pass has asked for before, and arm
        them: the next conv_weight_images / pw_weight_images call for such a weight takes its pair from the bank instead of launching.
        Call right before the forward pass of a training step (after the optimizer step that changed the weights).  A weight that is
        not armed -- not registered, first sighting, a second forward pass without a refresh, changed in place since (version
        counter) -- is split by its own launch as before: never stale."""
        self._bank().refresh()

    def weight_bank_invalidate(self):
        """Every armed image pair is stale from now on: call after the parameters were written behind torch's back (raw device pointers
        -- pvcnn_amd.optim.FlatAdam; in-place torch updates are caught by the tensors' version counters already)."""
        bank = getattr(self, '_weight_bank', None)
        if bank is not None:
            bank.epoch += 1

    def _bank(self):
        if getattr(self, '_weight_bank', None) is None:
            self._weight_bank = _WeightBank(self)
        return self._weight_bank

    def conv_weight_images(self, weight, nsplit):
        """(forward image, backward-data image) of a Conv3d weight; f16x2: both from ONE launch (a training step needs both and the
        weights do not change between its forward and its backward)."""
        co, ci = weight.shape[0], weight.shape[1]
        if int(nsplit) != 2:
            hit = self._bank().take('conv', weight, 1) if int(nsplit) == 1 else None      # plain bf16 (autocast): batched as well
            if hit is not None:
                return hit
            return self._conv_wsplit(weight, False, nsplit), self._conv_wsplit(weight, True, nsplit)
        hit = self._bank().take('conv', weight)
        if hit is not None:
            return hit
        wf = torch.empty((self.lib.pvcnn_conv3d_weight_split_bytes(co, ci, 0, 2),), dtype=torch.uint8, device=weight.device)
        wb = torch.empty((self.lib.pvcnn_conv3d_weight_split_bytes(co, ci, 1, 2),), dtype=torch.uint8, device=weight.device)
        with _Launch(weight) as s:
            _lib.check(self.lib.pvcnn_conv3d_weight_split_pair(_p(weight), co, ci, _p(wf), _p(wb), s), 'conv3d_weight_split_pair')
        return wf, wb

    def pw_weight_images(self, weight, nsplit):
        """(forward image, backward-data image) of a 1x1 convolution weight (Co, Ci); f16x2: one launch."""
        co, ci = weight.shape
        if int(nsplit) != 2:
            hit = self._bank().take('pw', weight, 1) if int(nsplit) == 1 else None
            if hit is not None:
                return hit
            return self._pw_wsplit(weight, False, nsplit), self._pw_wsplit(weight, True, nsplit)
        hit = self._bank().take('pw', weight)
        if hit is not None:
            return hit
        wf = torch.empty((self.lib.pvcnn_pwconv_weight_split_bytes(co, ci, 0, 2),), dtype=torch.uint8, device=weight.device)
        wb = torch.empty((self.lib.pvcnn_pwconv_weight_split_bytes(co, ci, 1, 2),), dtype=torch.uint8, device=weight.device)
        with _Launch(weight) as s:
            _lib.check(self.lib.pvcnn_pwconv_weight_split_pair(_p(weight), co, ci, _p(wf), _p(wb), s), 'pwconv_weight_split_pair')
        return wf, wb

    def conv3d_forward_split(self, x, weight, bias, nsplit, want_stats=False, amax=None):
        _f32(x, 'x'); _f32(weight, 'weight')
        _shape(x.dim() == 5 and weight.dim() == 5 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[1] == x.shape[1]
               and x.shape[2] == x.shape[3] == x.shape[4], 'conv3d: x (B,Ci,R,R,R), weight (Co,Ci,3,3,3) expected')
        if bias is not None:
            _f32(bias, 'bias')
        return self.conv3d_igemm_split(x, self._conv_wsplit(weight, False, nsplit), bias, weight.shape[0], nsplit, want_stats,
                                       amax if amax is not None else (self.conv_amax(x) if int(nsplit) == 2 else None))

    def conv3d_igemm_split(self, x, wts, bias, co, nsplit, want_stats=False, amax=None):
        """The implicit-GEMM launch alone (pre-split weight image `wts`; f16x2: `amax` = conv_amax(x), or a 1-word absmax_bits(x)
        for the single-scale mode): x (B,Ci,R,R,R) -> y (B,co,R,R,R) [, stats partials]."""
        b, ci, r = x.shape[0], x.shape[1], x.shape[2]
        y = torch.empty((b, co, r, r, r), dtype=torch.float32, device=x.device)
        part = None
        if want_stats:
            part = torch.empty((co, self.lib.pvcnn_conv3d_fwd_split_stats_parts(b, co, r, int(nsplit)), 2), dtype=torch.float32, device=x.device)
        if int(nsplit) == 2 and amax is None:
            amax = self.conv_amax(x)
        seg = self._amax_seg(amax, b * r * r, r) if int(nsplit) == 2 else 0
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_conv3d_fwd_split(_p(x), _p(wts), _p(bias) if bias is not None else None, b, ci, co, r, int(nsplit),
                                                       _p(amax) if amax is not None else None, seg,
                                                       _p(y), _p(part) if want_stats else None, s), 'conv3d_forward_split')
        return (y, part) if want_stats else y

    def conv3d_backward_data_split(self, grad_y, weight, nsplit, amax=None):
        _f32(grad_y, 'grad_y'); _f32(weight, 'weight')
        b, co, r = grad_y.shape[0], grad_y.shape[1], grad_y.shape[2]
        ci = weight.shape[1]
        # a convolution with Ci and Co exchanged on the flipped weights
        return self.conv3d_igemm_split(grad_y, self._conv_wsplit(weight, True, nsplit), None, ci, nsplit, False,
                                       amax if amax is not None else (self.conv_amax(grad_y) if int(nsplit) == 2 else None))

    # ---- backward-weight in f16x2 (csrc/conv3d_wgrad_f16.hip): R = 8, 12, 16 and 32; other grids stay on the fp32-MFMA kernel ----
    def conv3d_backward_weight_f16_serves(self, x):
        return x.dim() == 5 and x.shape[2] in (8, 12, 16, 32)

    def conv3d_backward_weight_f16(self, x, grad_y, x_amax=None, gy_amax=None, with_bias=False, out_w=None, out_b=None):
        """grad_w (Co,Ci,3,3,3) [, grad_bias]: x (B,Ci,R,R,R), grad_y (B,Co,R,R,R); *_amax = amax buffers of the two tensors (word [0],
        the global maximum, is what this kernel scales by)."""
        _f32(x, 'x'); _f32(grad_y, 'grad_y')
        b, ci, r = x.shape[0], x.shape[1], x.shape[2]
        co = grad_y.shape[1]
        _shape(self.conv3d_backward_weight_f16_serves(x) and tuple(grad_y.shape) == (b, co, r, r, r), 'conv3d_backward_weight_f16: R must be 8, 12, 16 or 32')
        x_amax = x_amax if x_amax is not None else self.absmax_bits(x)
        gy_amax = gy_amax if gy_amax is not None else self.absmax_bits(grad_y)
        gw = self._grad_out(out_w, (co, ci, 3, 3, 3), x.device)
        gb = self._grad_out(out_b, (co,), x.device) if with_bias else None
        ws = self._scratch(self.lib.pvcnn_conv3d_bwd_weight_f16_workspace_bytes(b, ci, co, r), x.device)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_conv3d_bwd_weight_f16(_p(x), _p(grad_y), _p(x_amax), self._amax_seg(x_amax, b * r * r, r), _p(gy_amax),
                                                            self._amax_seg(gy_amax, b * r * r, r), b, ci, co, r, _p(gw),
                                                            _p(gb) if with_bias else None, _p(ws), ws.numel(), s), 'conv3d_backward_weight_f16')
        return (gw, gb) if with_bias else gw

    # ---- SharedMLP 1x1 convolutions as channel-major MFMA GEMMs (csrc/pointwise.hip) --------------------
    has_pwconv = True

    def pwconv_forward(self, x, weight, bias, want_stats=False):
        """x (B,Ci,N), weight (Co,Ci), bias (Co) or None -> y (B,Co,N)."""
        _f32(x, 'x'); _f32(weight, 'weight')
        _shape(x.dim() == 3 and weight.dim() == 2 and weight.shape[1] == x.shape[1], 'pwconv: x (B,Ci,N), weight (Co,Ci) expected')
        if bias is not None:
            _f32(bias, 'bias')
        b, ci, n = x.shape
        co = weight.shape[0]
        k32 = (ci + 31) // 32 * 32                 # wt rows: K rounded up to the kernel's 32-channel chunk (zero tail)
        wt = torch.empty((k32, co), dtype=torch.float32, device=x.device)
        y = torch.empty((b, co, n), dtype=torch.float32, device=x.device)
        part = None
        if want_stats:
            part = torch.empty((co, self.lib.pvcnn_pwconv_fwd_stats_parts(b, n), 2), dtype=torch.float32, device=x.device)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_pwconv_transpose(_p(weight), co, ci, _p(wt), s), 'pwconv_transpose')
            if want_stats:
                _lib.check(self.lib.pvcnn_pwconv_fwd_stats(_p(x), _p(wt), k32, _p(bias) if bias is not None else None, b, ci, co, n,
                                                           _p(y), _p(part), s), 'pwconv_forward')
            else:
                _lib.check(self.lib.pvcnn_pwconv_fwd(_p(x), _p(wt), k32, _p(bias) if bias is not None else None, b, ci, co, n, _p(y), s), 'pwconv_forward')
        return (y, part) if want_stats else y

    def pwconv_backward_data(self, grad_y, weight):
        """grad_y (B,Co,N), weight (Co,Ci) -> grad_x (B,Ci,N): the forward GEMM with K = Co on the weight as stored."""
        _f32(grad_y, 'grad_y'); _f32(weight, 'weight')
        b, co, n = grad_y.shape
        ci = weight.shape[1]
        gx = torch.empty((b, ci, n), dtype=torch.float32, device=grad_y.device)
        with _Launch(grad_y) as s:
            _lib.check(self.lib.pvcnn_pwconv_fwd(_p(grad_y), _p(weight), co, None, b, co, ci, n, _p(gx), s), 'pwconv_backward_data')
        return gx

    # ---- the same GEMMs on the bf16 matrix cores (csrc/pointwise_bf16.hip), nsplit = 3 (bf16x3) or 1 (bf16) ----
    has_pwconv_split = True
    # arithmetic of the SharedMLP forward / backward-data products: 'fp32' (pointwise.hip), 'f16x2' or 'bf16x3' (pointwise_bf16.hip)
    pw_math = os.environ.get('PVCNN_PW_MATH', 'f16x2')
    PW_NSPLIT = {'f16x2': 2, 'bf16x3': 3, 'fp32': 0}
    # K * M * B * N below which the split forward / backward-data path is not worth its extra launches (weight images, an amax pass
    # when the producer left no table).  Round 3, after the straight-line / pipelined kernels: a sweep of the threshold from 2^32
    # (round 2) down to "always" on one box -- PVCNN 2121 -> 2157, PVCNN++ 429 -> 484, ShapeNet 1418 -> 1497 clouds/s, Frustum
    # 3763 -> 4095 frustums/s, flat below 2^24 (profiles/ab/r03w_*).  Backward-weight has its own bar: the f16x2 kernel writes
    # 128 x 128 partial tiles per partition of the points, a loss on small weight matrices (64 x 64: 0.17 vs 0.05 ms).
    pw_split_min_macs = 1 << 24
    pw_wgrad_f16_min_macs = 1 << 32

    def _pw_wsplit(self, weight, for_bwd_data, nsplit):
        co, ci = weight.shape
        nbytes = self.lib.pvcnn_pwconv_weight_split_bytes(co, ci, int(for_bwd_data), int(nsplit))
        wts = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
        with _Launch(weight) as s:
            _lib.check(self.lib.pvcnn_pwconv_weight_split(_p(weight), co, ci, int(for_bwd_data), int(nsplit), _p(wts), s), 'pwconv_weight_split')
        return wts

    def pwconv_gemm_split(self, x, wts, bias, m, nsplit, want_stats=False, amax=None):
        """The GEMM launch alone: x (B,K,N), pre-split weight image (f16x2: amax = pw_amax(x), or a 1-word absmax_bits(x) for the
        single-scale mode) -> y (B,m,N) [, stats partials]."""
        b, k, n = x.shape
        y = torch.empty((b, m, n), dtype=torch.float32, device=x.device)
        part = None
        if want_stats:
            part = torch.empty((m, self.lib.pvcnn_pwconv_fwd_split_stats_parts(b, n), 2), dtype=torch.float32, device=x.device)
        if int(nsplit) == 2 and amax is None:
            amax = self.pw_amax(x)
        seg = self._amax_seg(amax, b * ((n + self.PW_AMAX_SEG - 1) // self.PW_AMAX_SEG), self.PW_AMAX_SEG) if int(nsplit) == 2 else 0
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_pwconv_fwd_split(_p(x), _p(wts), _p(bias) if bias is not None else None, b, k, m, n, int(nsplit),
                                                       _p(amax) if amax is not None else None, seg,
                                                       _p(y), _p(part) if want_stats else None, s), 'pwconv_forward_split')
        return (y, part) if want_stats else y

    def pwconv_forward_split(self, x, weight, bias, nsplit, want_stats=False, amax=None):
        _f32(x, 'x'); _f32(weight, 'weight')
        _shape(x.dim() == 3 and weight.dim() == 2 and weight.shape[1] == x.shape[1], 'pwconv: x (B,Ci,N), weight (Co,Ci) expected')
        if bias is not None:
            _f32(bias, 'bias')
        return self.pwconv_gemm_split(x, self._pw_wsplit(weight, False, nsplit), bias, weight.shape[0], nsplit, want_stats, amax)

    def pwconv_backward_data_split(self, grad_y, weight, nsplit, amax=None):
        _f32(grad_y, 'grad_y'); _f32(weight, 'weight')
        return self.pwconv_gemm_split(grad_y, self._pw_wsplit(weight, True, nsplit), None, weight.shape[1], nsplit, False, amax)

    def pwconv_backward_weight(self, x, grad_y, with_bias=False, out_w=None, out_b=None):
        """-> grad_weight (Co,Ci), or (grad_weight, grad_bias) when with_bias."""
        _f32(x, 'x'); _f32(grad_y, 'grad_y')
        b, ci, n = x.shape
        co = grad_y.shape[1]
        gw = self._grad_out(out_w, (co, ci), x.device)
        gb = self._grad_out(out_b, (co,), x.device) if with_bias else None
        ws = self._scratch(self.lib.pvcnn_pwconv_bwd_weight_workspace_bytes(b, ci, co, n), x.device)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_pwconv_bwd_weight(_p(x), _p(grad_y), b, ci, co, n, _p(gw), _p(gb) if with_bias else None,
                                                        _p(ws), ws.numel(), s), 'pwconv_backward_weight')
        return (gw, gb) if with_bias else gw

    def pwconv_backward_weight_f16_serves(self, x):
        return x.dim() == 3 and x.shape[2] % 4 == 0

    def pwconv_backward_weight_f16(self, x, grad_y, x_amax=None, gy_amax=None, with_bias=False, out_w=None, out_b=None):
        """f16x2 on the fp16 matrix cores (csrc/pointwise_wgrad_f16.hip): -> grad_weight (Co,Ci) [, grad_bias]."""
        _f32(x, 'x'); _f32(grad_y, 'grad_y')
        b, ci, n = x.shape
        co = grad_y.shape[1]
        _shape(self.pwconv_backward_weight_f16_serves(x) and tuple(grad_y.shape) == (b, co, n), 'pwconv_backward_weight_f16: N must be a multiple of 4')
        x_amax = x_amax if x_amax is not None else self.absmax_bits(x)
        gy_amax = gy_amax if gy_amax is not None else self.absmax_bits(grad_y)
        gw = self._grad_out(out_w, (co, ci), x.device)
        gb = self._grad_out(out_b, (co,), x.device) if with_bias else None
        ws = self._scratch(self.lib.pvcnn_pwconv_bwd_weight_f16_workspace_bytes(b, ci, co, n), x.device)
        with _Launch(x) as s:
            tiles = b * ((n + self.PW_AMAX_SEG - 1) // self.PW_AMAX_SEG)
            _lib.check(self.lib.pvcnn_pwconv_bwd_weight_f16(_p(x), _p(grad_y), _p(x_amax), self._amax_seg(x_amax, tiles, self.PW_AMAX_SEG), _p(gy_amax),
                                                            self._amax_seg(gy_amax, tiles, self.PW_AMAX_SEG), b, ci, co, n, _p(gw),
                                                            _p(gb) if with_bias else None, _p(ws), ws.numel(), s), 'pwconv_backward_weight_f16')
        return (gw, gb) if with_bias else gw

    # ---- BatchNorm + ReLU/LeakyReLU in two passes each way (csrc/bnact.hip) ---------------------------
    has_bnact = True

    BNACT_AMAX_MAX_SEG = 256   # the apply passes emit amax buffers for segments up to this long

    has_bnact_dropout = True

    def dropout_keep_mask(self, seed, p, numel):
        """keep(e), e = 0 .. numel - 1, of the dropout fused into bnact_forward / bnact_backward (drop=(seed, p)) -> bool tensor (tests)."""
        keep = torch.empty((int(numel),), dtype=torch.uint8, device=seed.device)
        with _Launch(seed) as s:
            _lib.check(self.lib.pvcnn_dropout_keep_mask(_p(seed), float(p), int(numel), _p(keep), s), 'dropout_keep_mask')
        return keep.bool()

    def bnact_forward(self, x, gamma, beta, running_mean, running_var, training, momentum, eps, slope, stats=None, amax_seg=0,
                      y_amax=None, drop=None):
        """x (B,C,S) -> (y, mean, rstd).  Training: batch statistics (running stats updated in place);
        eval: running statistics.  stats = (mean, rstd) already known (from a convolution epilogue +
        bn_finalize): only the normalise + activate pass runs.
        amax_seg > 0: -> (y, mean, rstd, y_amax), y's amax buffer with segments of amax_seg positions emitted by the apply pass.
        y_amax given: the buffer bn_finalize(..., zero_word=y_amax) already armed (all of it is zero: the pass then needs no
        reduction launch behind it).
        drop = (seed: one int64 on the device, p): the nn.Dropout(p) behind the pair, applied by the same pass (needs amax_seg > 0)."""
        _f32(x, 'x')
        b, c, s3 = x.shape
        dev = x.device
        y = torch.empty_like(x)
        if stats is not None:
            mean, rstd = stats
            training = False                    # the kernel entry's "statistics are given" mode
        elif training:
            mean = torch.empty((c,), dtype=torch.float32, device=dev)
            rstd = torch.empty((c,), dtype=torch.float32, device=dev)
        else:
            mean = running_mean.clone()          # saved for backward: must not alias the live buffer
            rstd = torch.rsqrt(running_var + eps)
        ws = self._scratch(self.lib.pvcnn_bnact_workspace_bytes(b, c, s3), dev)
        nul = ctypes.c_void_p(None)
        amax_seg = int(amax_seg)
        armed = y_amax is not None
        if amax_seg > 0 and not armed:
            y_amax = self.amax_buffer(b, s3, amax_seg, dev)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_bnact_fwd(_p(x), _p(gamma) if gamma is not None else nul, _p(beta) if beta is not None else nul,
                                                _p(running_mean) if (training and running_mean is not None) else nul,
                                                _p(running_var) if (training and running_var is not None) else nul,
                                                b, c, s3, float(eps), float(momentum), float(slope), int(bool(training)),
                                                _p(mean), _p(rstd), _p(y), _p(y_amax) if amax_seg > 0 else nul, amax_seg, int(armed),
                                                _p(ws), ws.numel(), _p(drop[0]) if drop else nul, float(drop[1]) if drop else 0.0, s),
                       'bnact_forward')
        return (y, mean, rstd, y_amax) if amax_seg > 0 else (y, mean, rstd)

    has_bnact_rowmax = True

    def amax_and_row_keys(self, b, c, n, seg, device):
        """One uninitialised int32 buffer holding the amax buffer of a (b, c, n) tensor AND its b * c 64-bit row keys (8-byte aligned)
        -> (whole buffer: hand it to bn_finalize(zero_word=...), amax view, keys view)."""
        words = self.lib.pvcnn_absmax_tiles_count(int(b), int(n), int(seg))
        off = (words + 1) // 2 * 2
        whole = torch.empty((off + 2 * int(b) * int(c),), dtype=torch.int32, device=device)
        return whole, whole[:words], whole[off:]

    def bnact_apply_rowmax(self, x, gamma, beta, mean, rstd, slope, amax_seg, y_amax, row_keys, out=None):
        """The apply pass of bnact_forward on known statistics, also emitting the row maxima of y: x (B,C,S) -> (y, winners (B,C) int64,
        values (B,C)) == (y, *y.max(dim=-1)[::-1]).  y_amax / row_keys: the views of amax_and_row_keys, ZEROED (bn_finalize)."""
        _f32(x, 'x')
        b, c, s3 = x.shape
        _shape(s3 % 256 == 0 and amax_seg % 4 == 0 and 0 < amax_seg <= 256 and 256 % amax_seg == 0,
               'bnact_apply_rowmax: S % 256 == 0 and an amax_seg that is a multiple of 4 and divides 256 expected')
        # out: where y goes -- a (B, C, S) view whose rows are contiguous inside a sample and whose samples may be further apart: the
        # channel slice of the classifier's concatenation that this tensor will be (the concatenation then copies nothing for it)
        if out is None:
            y = torch.empty_like(x)
        else:
            y = out
            _dev(y, 'out')
            _shape(tuple(y.shape) == tuple(x.shape) and y.dtype == torch.float32 and y.stride(2) == 1 and y.stride(1) == s3
                   and y.stride(0) >= c * s3 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0,
                   'bnact_apply_rowmax: out must be a (B, C, S) float32 view with contiguous, 16-byte aligned rows')
        ybs = int(y.stride(0)) if b > 1 else c * s3
        winners = torch.empty((b, c), dtype=torch.int64, device=x.device)
        values = torch.empty((b, c), dtype=torch.float32, device=x.device)
        nul = ctypes.c_void_p(None)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_bnact_apply_rowmax(_p(x), _p(gamma) if gamma is not None else nul, _p(beta) if beta is not None else nul,
                                                         _p(mean), _p(rstd), b, c, s3, float(slope), _p(y), ybs, _p(y_amax), int(amax_seg),
                                                         _p(row_keys), s), 'bnact_apply_rowmax')
            _lib.check(self.lib.pvcnn_row_keys_decode(_p(row_keys), _p(y), b * c, s3, _p(winners), _p(values), c, ybs, s), 'row_keys_decode')
        return y, winners, values

    has_devox_bnact = True

    def amax_buffer(self, b, n, seg, device):
        """Uninitialised amax buffer for a (b, C, n) tensor with segments of `seg` positions."""
        return torch.empty((self.lib.pvcnn_absmax_tiles_count(int(b), int(n), int(seg)),), dtype=torch.int32, device=device)

    def bn_finalize(self, part, count, running_mean, running_var, momentum, eps, shift=None, zero_word=None, counter=None):
        """(C, nparts, 2) partial sums of (y - shift) from a convolution epilogue (shift = that convolution's bias, or None)
        -> (mean, rstd) of y; running stats updated in place.  zero_word: an amax buffer this launch zeroes (arming it for the
        apply pass that follows, which fills it by atomic maxima: bnact_forward(..., y_amax=zero_word)).  counter: the module's int64 num_batches_tracked,
        incremented by the same launch."""
        c, nparts = part.shape[0], part.shape[1]
        dev = part.device
        mean = torch.empty((c,), dtype=torch.float32, device=dev)
        rstd = torch.empty((c,), dtype=torch.float32, device=dev)
        nul = ctypes.c_void_p(None)
        with _Launch(part) as s:
            _lib.check(self.lib.pvcnn_bn_finalize(_p(part), c, nparts, float(count), float(eps), float(momentum),
                                                  _p(shift) if shift is not None else nul,
                                                  _p(running_mean) if running_mean is not None else nul,
                                                  _p(running_var) if running_var is not None else nul, _p(mean), _p(rstd),
                                                  _p(zero_word) if zero_word is not None else nul,
                                                  zero_word.numel() if zero_word is not None else 0,
                                                  _p(counter) if counter is not None else nul, s),
                       'bn_finalize')
        return mean, rstd

    def bn_stats(self, x, running_mean, running_var, momentum, eps):
        """Training-mode statistics of x (B,C,S): -> (mean, rstd); running stats updated in place (may be None)."""
        _f32(x, 'x')
        b, c, s3 = x.shape
        dev = x.device
        mean = torch.empty((c,), dtype=torch.float32, device=dev)
        rstd = torch.empty((c,), dtype=torch.float32, device=dev)
        ws = self._scratch(self.lib.pvcnn_bnact_workspace_bytes(b, c, s3), dev)
        nul = ctypes.c_void_p(None)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_bn_stats(_p(x), _p(running_mean) if running_mean is not None else nul,
                                               _p(running_var) if running_var is not None else nul, b, c, s3, float(eps),
                                               float(momentum), _p(mean), _p(rstd), _p(ws), ws.numel(), s), 'bn_stats')
        return mean, rstd

    def trilinear_devoxelize_bnact_forward(self, r, is_training, coords, features, gamma, beta, mean, rstd, slope, addend=None,
                                           se_scale=None):
        """trilinear_devoxelize_forward of leaky_relu(bn(features)) [* se_scale (B,C): SE3d's excitation] without materialising that
        tensor: features (B,C,R^3) is the PRE-BatchNorm grid, mean / rstd (C) its statistics."""
        _f32(features, 'features'); _f32(coords, 'coords')
        r = int(r)
        _shape(features.dim() == 3 and coords.dim() == 3 and coords.shape[1] == 3
               and coords.shape[0] == features.shape[0] and features.shape[2] == r * r * r,
               'trilinear_devoxelize: coords (B,3,N), features (B,C,R^3) expected')
        b, c = features.shape[:2]
        n = coords.shape[2]
        dev = features.device
        if addend is not None:
            _f32(addend, 'addend')
            _shape(tuple(addend.shape) == (b, c, n), 'trilinear_devoxelize: addend (B,C,N) expected')
        if se_scale is not None:
            _f32(se_scale, 'se_scale')
            _shape(tuple(se_scale.shape) == (b, c), 'trilinear_devoxelize: se_scale (B,C) expected')
        outs = torch.empty((b, c, n), dtype=torch.float32, device=dev)
        if is_training:
            inds = torch.empty((b, 8, n), dtype=torch.int32, device=dev)
            wgts = torch.empty((b, 8, n), dtype=torch.float32, device=dev)
        else:
            inds, wgts = _dummies(dev)
        nul = ctypes.c_void_p(None)
        with _Launch(features) as s:
            _lib.check(self.lib.pvcnn_trilinear_devox_bnact_fwd(
                _p(coords), _p(features), _p(gamma) if gamma is not None else nul, _p(beta) if beta is not None else nul,
                _p(mean), _p(rstd), float(slope), b, c, n, r, int(bool(is_training)),
                _p(inds) if is_training else None, _p(wgts) if is_training else None,
                _p(addend) if addend is not None else None, _p(se_scale) if se_scale is not None else None, _p(outs), s),
                'trilinear_devoxelize_bnact_forward')
        return [outs, inds, wgts]

    # ---- torch.cat(features, dim=1) of the classifier input + the amax buffer of its output in one pass (csrc/bnact.hip) ----
    has_concat_points = True

    def concat_points(self, tensors, want_amax=True, out=None, in_place=None, want_global=True):
        """tensors: (B, C_i, N) float32, each with contiguous rows inside a cloud (a channel slice is fine) or broadcast over the points
        (stride 0 along N, contiguous (B, C_i)) -> (out (B, sum C_i, N), its amax buffer with 256-point segments | None).
        out: the (B, sum C_i, N) buffer to fill; in_place = {i: amax buffer of tensors[i]}: those sources ALREADY ARE their channel slice
        of `out` (the pass that produced them wrote them there: bnact_apply_rowmax(..., out=)): nothing is copied for them.
        want_global=False: the amax buffer's table only, no word [0] (see absmax_tiles)."""
        _shape(0 < len(tensors) <= 8, 'concat_points: 1..8 sources')
        b, n = tensors[0].shape[0], tensors[0].shape[2]
        ptrs, bstr, chans, pstr = [], [], [], []
        for i, t in enumerate(tensors):
            _dev(t, 'source')
            _shape(t.dim() == 3 and t.dtype == torch.float32 and t.shape[0] == b and t.shape[2] == n, 'concat_points: (B, C_i, N) float sources expected')
            c = t.shape[1]
            if n > 1 and t.stride(2) == 0:
                _shape(t.stride(1) == 1 or c == 1, 'concat_points: a broadcast source must be contiguous over (B, C)')
                pstr.append(0); bstr.append(t.stride(0) if b > 1 else c)
            else:
                _shape((n == 1 or t.stride(2) == 1) and (c == 1 or t.stride(1) == n), 'concat_points: rows must be contiguous inside a cloud')
                # (a single cloud: torch leaves stride(0) of a size-1 dimension arbitrary, so it is stated -- except for a source that
                #  already IS its channel slice of `out`, whose cloud stride is the buffer's and is checked as such by the library)
                pstr.append(1); bstr.append(t.stride(0) if (b > 1 or (in_place and i in in_place)) else c * n)
            ptrs.append(t.data_ptr()); chans.append(c)
        k = len(tensors)
        if out is None:
            out = torch.empty((b, sum(chans), n), dtype=torch.float32, device=tensors[0].device)
        else:
            _f32(out, 'out')
            _shape(tuple(out.shape) == (b, sum(chans), n), 'concat_points: out (B, sum C_i, N) expected')
        pre = [None] * k
        for i, table in (in_place or {}).items():
            _shape(want_amax and table is not None and table.dtype == torch.int32 and table.is_cuda, 'concat_points: an in-place source comes with its amax buffer')
            pre[i] = table.data_ptr()
        amax = self.amax_buffer(b, n, self.PW_AMAX_SEG, out.device) if want_amax else None
        ticket = self._tickets(1, out.device) if want_amax else None
        with _Launch(out) as s:
            _lib.check(self.lib.pvcnn_concat_points((ctypes.c_void_p * k)(*ptrs), (ctypes.c_long * k)(*bstr), (ctypes.c_int * k)(*chans),
                                                    (ctypes.c_int * k)(*pstr), (ctypes.c_void_p * k)(*pre) if in_place else None, k, b, n,
                                                    _p(out), _p(amax) if want_amax else None,
                                                    (_p(ticket) if ticket is not None else None) if want_global or self.amax_global or not want_amax else self._TABLE_ONLY,
                                                    s), 'concat_points')
        return out, amax

    # ---- the two halves of bnact_backward on their own (PVConv's SE tail puts the excitation's backward between them) ----
    has_bnact_split_bwd = True

    # ---- Linear + BatchNorm1d + ReLU on a handful of rows (models/utils.py:11-12: the cloud-descriptor heads), csrc/dense.hip ----
    has_dense_bn_relu = True

    def dense_bn_relu_supported(self, rows, cin, cout):
        return bool(self.lib.pvcnn_dense_bn_relu_supported(int(rows), int(cin), int(cout)))

    def dense_bn_relu_forward(self, x, weight, bias, gamma, beta, running_mean, running_var, counter, eps, momentum):
        """x (rows, Cin) -> (y (rows, Cout), z, mean, rstd): y = relu(batch_norm(x W^T + bias)) with batch statistics in ONE launch;
        running statistics / num_batches_tracked updated in place (None: not tracked)."""
        _f32(x, 'x'); _f32(weight, 'weight')
        _shape(x.dim() == 2 and weight.dim() == 2 and weight.shape[1] == x.shape[1], 'dense_bn_relu: x (rows, Cin), weight (Cout, Cin) expected')
        rows, cin = x.shape
        cout = weight.shape[0]
        dev = x.device
        z = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        y = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        mean = torch.empty((cout,), dtype=torch.float32, device=dev)
        rstd = torch.empty((cout,), dtype=torch.float32, device=dev)
        opt = lambda t: _p(t) if t is not None else ctypes.c_void_p(None)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_dense_bn_relu_fwd(_p(x), _p(weight), opt(bias), opt(gamma), opt(beta), opt(running_mean), opt(running_var),
                                                        opt(counter), rows, cin, cout, float(eps), float(momentum), _p(z), _p(y), _p(mean),
                                                        _p(rstd), s), 'dense_bn_relu_forward')
        return y, z, mean, rstd

    def dense_bn_relu_backward(self, x, grad_y, z, mean, rstd, gamma, beta, out_w=None, out_b=None, out_gamma=None, out_beta=None):
        """-> (grad_z (rows, Cout), grad_weight (Cout, Cin), grad_bias, grad_gamma, grad_beta) in ONE launch (grad_x = grad_z @ weight is
        the caller's GEMM).  out_*: where to write the parameter gradients (slots of a flat gradient bucket), else fresh tensors."""
        _f32(x, 'x'); _f32(grad_y, 'grad_y')
        rows, cin = x.shape
        cout = z.shape[1]
        dev = x.device
        gz = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        gw = self._grad_out(out_w, (cout, cin), dev)
        gb = self._grad_out(out_b, (cout,), dev)
        gg = self._grad_out(out_gamma, (cout,), dev)
        gbeta = self._grad_out(out_beta, (cout,), dev)
        opt = lambda t: _p(t) if t is not None else ctypes.c_void_p(None)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_dense_bn_relu_bwd(_p(x), _p(grad_y), _p(z), _p(mean), _p(rstd), opt(gamma), opt(beta), rows, cin, cout,
                                                        _p(gz), _p(gw), _p(gb), _p(gg), _p(gbeta), s), 'dense_bn_relu_backward')
        return gz, gw, gb, gg, gbeta

    # ---- max over the neighbours of a centre (modules/pointnet.py:85) ------------------------------------------------
    has_neighbor_max = True

    def neighbor_max_supported(self, k):
        return bool(self.lib.pvcnn_neighbor_max_supported(int(k)))

    def neighbor_max_forward(self, x):
        """x (..., K) contiguous -> (max over K (...), winners (...) uint8)."""
        _f32(x, 'x')
        k = x.shape[-1]
        rows = x.numel() // k
        out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        winners = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_neighbor_max_fwd(_p(x), rows, k, _p(out), _p(winners), s), 'neighbor_max_forward')
        return out, winners

    def neighbor_max_backward(self, grad_out, winners, k):
        """-> grad_x (..., K): grad_out at the winner, zeros elsewhere (one write pass)."""
        _f32(grad_out, 'grad_out')
        _shape(grad_out.shape == winners.shape and winners.dtype == torch.uint8 and winners.is_contiguous(), 'neighbor_max: winners (...) uint8 expected')
        gx = torch.empty(tuple(grad_out.shape) + (int(k),), dtype=torch.float32, device=grad_out.device)
        with _Launch(grad_out) as s:
            _lib.check(self.lib.pvcnn_neighbor_max_bwd(_p(grad_out), _p(winners), grad_out.numel(), int(k), _p(gx), s), 'neighbor_max_backward')
        return gx

    def row_argmax(self, x, with_values=False):
        """x (..., K) contiguous, K % 4 == 0 -> winners (...) int64 [, values (...)]: `x.max(dim=-1)` for long rows, one read."""
        _f32(x, 'x')
        k = x.shape[-1]
        _shape(k % 4 == 0 and k > 0, 'row_argmax: the last dimension must be a multiple of 4')
        winners = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
        values = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device) if with_values else None
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_row_argmax(_p(x), x.numel() // k, k, _p(winners), _p(values), s), 'row_argmax')
        return (winners, values) if with_values else winners

    has_frustum_loss = True

    def frustum_box_loss(self, center, center_reg, heading_scores, size_scores, hrn, srn, hr, sr, heading_bin_id, size_template_id,
                         heading_residual, size_residual, center_t, templates, bin_centers, bin_width, w_heading, w_size, w_corners):
        """csrc/frustum.hip: the box part of FrustumPointNetLoss (modules/frustum.py:43-124) and its gradient in one launch ->
        (loss, 0-dim; grads, flat: the gradients with respect to the first eight arguments, concatenated in that order)."""
        outs = (center, center_reg, heading_scores, size_scores, hrn, srn, hr, sr)
        for t in outs + (heading_residual, size_residual, center_t, templates, bin_centers):
            _f32(t, 'frustum_box_loss input')
        b, nh, ns = center.shape[0], heading_scores.shape[1], size_scores.shape[1]
        _shape(tuple(center.shape) == (b, 3) and tuple(center_reg.shape) == (b, 3) and tuple(center_t.shape) == (b, 3)
               and tuple(heading_scores.shape) == (b, nh) and tuple(hrn.shape) == (b, nh) and tuple(hr.shape) == (b, nh)
               and tuple(size_scores.shape) == (b, ns) and tuple(srn.shape) == (b, ns, 3) and tuple(sr.shape) == (b, ns, 3)
               and tuple(heading_residual.shape) == (b,) and tuple(size_residual.shape) == (b, 3) and tuple(templates.shape) == (ns, 3)
               and tuple(bin_centers.shape) == (nh,) and tuple(heading_bin_id.shape) == (b,) and tuple(size_template_id.shape) == (b,)
               and heading_bin_id.dtype == torch.int64 and size_template_id.dtype == torch.int64
               and heading_bin_id.is_contiguous() and size_template_id.is_contiguous(), 'frustum_box_loss: shapes of modules/frustum.py:57-75 expected')
        loss = torch.empty((), dtype=torch.float32, device=center.device)
        grads = torch.empty((self.lib.pvcnn_frustum_box_loss_grad_floats(b, nh, ns),), dtype=torch.float32, device=center.device)
        with _Launch(center) as s:
            _lib.check(self.lib.pvcnn_frustum_box_loss(_p(center), _p(center_reg), _p(heading_scores), _p(size_scores), _p(hrn), _p(srn), _p(hr),
                                                       _p(sr), _p(heading_bin_id), _p(size_template_id), _p(heading_residual),
                                                       _p(size_residual), _p(center_t), _p(templates), _p(bin_centers), b, nh, ns,
                                                       float(bin_width), float(w_heading), float(w_size), float(w_corners), _p(loss),
                                                       _p(grads), s), 'frustum_box_loss')
        return loss, grads

    has_se_excite = True

    def se_excite_forward(self, part, gamma, beta, w1, w2, s3):
        """part (C,B,slices,2) from bnact_partial_sums_raw(grad_y=None) -> (a_sum (B,C), ax_sum (B,C), squeezed (B,C), hidden (B,H),
        excite (B,C)): the slice sums + SE3d's two Linear layers + ReLU + Sigmoid on the squeeze, one launch."""
        c, b, slices, _ = part.shape
        h = w1.shape[0]
        _shape(tuple(w1.shape) == (h, c) and tuple(w2.shape) == (c, h) and c <= 2048 and h <= 256, 'se_excite: W1 (H,C), W2 (C,H), C <= 2048, H <= 256')
        dev = part.device
        a_sum, ax_sum, squeezed, excite = (torch.empty((b, c), dtype=torch.float32, device=dev) for _ in range(4))
        hidden = torch.empty((b, h), dtype=torch.float32, device=dev)
        nul = ctypes.c_void_p(None)
        with _Launch(part) as s:
            _lib.check(self.lib.pvcnn_se_excite_fwd(_p(part), slices, _p(gamma) if gamma is not None else nul, _p(beta) if beta is not None else nul,
                                                    _p(w1), _p(w2), b, c, h, 1.0 / float(s3), _p(a_sum), _p(ax_sum), _p(squeezed), _p(hidden),
                                                    _p(excite), s), 'se_excite_fwd')
        return a_sum, ax_sum, squeezed, hidden, excite

    def se_excite_backward(self, part, a_sum, ax_sum, gamma, beta, squeezed, hidden, excite, w1, w2, s3):
        """part (C,B,slices,2) from bnact_partial_sums_raw(grad_y) -> (g_w1 (H,C), g_w2 (C,H), g_mean (B,C), sum_beta (C), sum_gamma (C)),
        two launches."""
        c, b, slices, _ = part.shape
        h = w1.shape[0]
        dev = part.device
        g_w1 = torch.empty((h, c), dtype=torch.float32, device=dev)
        g_w2 = torch.empty((c, h), dtype=torch.float32, device=dev)
        g_mean = torch.empty((b, c), dtype=torch.float32, device=dev)
        sum_beta = torch.empty((c,), dtype=torch.float32, device=dev)
        sum_gamma = torch.empty((c,), dtype=torch.float32, device=dev)
        ws = torch.empty((b * (3 * c + h),), dtype=torch.float32, device=dev)
        nul = ctypes.c_void_p(None)
        with _Launch(part) as s:
            _lib.check(self.lib.pvcnn_se_excite_bwd(_p(part), slices, _p(a_sum), _p(ax_sum), _p(gamma) if gamma is not None else nul,
                                                    _p(beta) if beta is not None else nul, _p(squeezed), _p(hidden), _p(excite), _p(w1), _p(w2),
                                                    b, c, h, 1.0 / float(s3), _p(g_w1), _p(g_w2), _p(g_mean), _p(sum_beta), _p(sum_gamma),
                                                    _p(ws), s), 'se_excite_bwd')
        return g_w1, g_w2, g_mean, sum_beta, sum_gamma

    def bnact_partial_sums_raw(self, x, grad_y, gamma, beta, mean, rstd, slope):
        """-> part (C,B,slices,2): per slice of positions the sums of g' and g' * xhat, g' = grad_y * act'(z) (grad_y None: == 1)."""
        _f32(x, 'x')
        b, c, s3 = x.shape
        gy_bstride = _f32_rows(grad_y, 'grad_y') if grad_y is not None else c * s3
        slices = self.lib.pvcnn_bnact_slices(s3)
        part = torch.empty((c, b, slices, 2), dtype=torch.float32, device=x.device)
        nul = ctypes.c_void_p(None)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_bnact_partial_sums(_p(x), _p(grad_y) if grad_y is not None else nul, gy_bstride,
                                                         _p(gamma) if gamma is not None else nul, _p(beta) if beta is not None else nul,
                                                         _p(mean), _p(rstd), b, c, s3, float(slope), _p(part), s), 'bnact_partial_sums')
        return part

    def bnact_partial_sums(self, x, grad_y, gamma, beta, mean, rstd, slope):
        """-> (P, Q) (B,C) each: sums over the positions of g' and g' * xhat, g' = grad_y * act'(z) (grad_y None: == 1)."""
        sums = self.bnact_partial_sums_raw(x, grad_y, gamma, beta, mean, rstd, slope).sum(dim=2)   # (C, B, 2)
        return sums[..., 0].t().contiguous(), sums[..., 1].t().contiguous()

    def bnact_backward_apply(self, x, grad_y, gamma, beta, mean, rstd, sum_gamma, sum_beta, slope, training, bc_mul=None, bc_add=None,
                             amax_seg=256):
        """grad_x of BatchNorm + activation with the two per-channel sums GIVEN and g' = (grad_y * bc_mul[b][c] + bc_add[b][c]) * act'(z);
        -> (grad_x, its amax buffer with segments of amax_seg positions)."""
        _f32(x, 'x')
        gy_bstride = _f32_rows(grad_y, 'grad_y')
        b, c, s3 = x.shape
        for t, name in ((bc_mul, 'bc_mul'), (bc_add, 'bc_add')):
            if t is not None:
                _f32(t, name)
                _shape(tuple(t.shape) == (b, c), f'{name} (B,C) expected')
        gx = torch.empty_like(x)
        amax = self.amax_buffer(b, s3, amax_seg, x.device)
        nul = ctypes.c_void_p(None)
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_bnact_bwd_apply(_p(x), _p(grad_y), gy_bstride, _p(gamma) if gamma is not None else nul,
                                                      _p(beta) if beta is not None else nul, _p(mean), _p(rstd),
                                                      _p(sum_gamma) if sum_gamma is not None else nul, _p(sum_beta) if sum_beta is not None else nul,
                                                      _p(bc_mul) if bc_mul is not None else nul, _p(bc_add) if bc_add is not None else nul,
                                                      b, c, s3, float(slope), int(bool(training)), _p(gx), _p(amax), int(amax_seg), s),
                       'bnact_backward_apply')
        return gx, amax

    def bnact_backward(self, x, grad_y, gamma, beta, mean, rstd, slope, training, amax_seg=0, drop=None, out_w=None, out_b=None):
        """-> (grad_x, grad_gamma, grad_beta [, grad_x's amax buffer with segments of amax_seg positions, emitted by the apply pass]).
        drop = (seed, p) of the forward call: grad_y is then the gradient of the DROPPED output (needs amax_seg > 0)."""
        _f32(x, 'x')
        gy_bstride = _f32_rows(grad_y, 'grad_y')
        b, c, s3 = x.shape
        dev = x.device
        gx = torch.empty_like(x)
        gg = self._grad_out(out_w, (c,), dev)
        gb = self._grad_out(out_b, (c,), dev)
        ws = self._scratch(self.lib.pvcnn_bnact_workspace_bytes(b, c, s3), dev)
        nul = ctypes.c_void_p(None)
        amax_seg = int(amax_seg)
        gx_amax = self.amax_buffer(b, s3, amax_seg, dev) if amax_seg > 0 else None
        tickets = self._tickets(c, dev)                   # one word per channel: the reduce pass finalises its own sums
        with _Launch(x) as s:
            _lib.check(self.lib.pvcnn_bnact_bwd_strided(_p(x), _p(grad_y), gy_bstride, _p(gamma) if gamma is not None else nul,
                                                        _p(beta) if beta is not None else nul, _p(mean), _p(rstd), b, c, s3, float(slope),
                                                        int(bool(training)), _p(gx), _p(gg), _p(gb),
                                                        _p(gx_amax) if amax_seg > 0 else nul, amax_seg, _p(ws), ws.numel(),
                                                        _p(drop[0]) if drop else nul, float(drop[1]) if drop else 0.0,
                                                        _p(tickets) if tickets is not None else nul, s), 'bnact_backward')
        return (gx, gg, gb, gx_amax) if amax_seg > 0 else (gx, gg, gb)


class _WeightBank:
    """Persistent f16x2 image pairs of registered weights (HipBackend.weight_bank_*).  An entry is keyed by (kind, data pointer,
    (Co, Ci)); `wanted` = the keys a forward pass asked for (take() misses note them), so a refresh computes what is used and nothing
    else; the device tables are rebuilt when that set changes (the warm-up steps), not in steady state."""

    def __init__(self, be):
        import weakref
        self.be, self._weakref = be, weakref
        self.params = []            # weakrefs of registered parameters
        self.seen = set()           # id()s of registered parameters
        self.wanted = set()
        self.entries = {}           # key -> dict(param=weakref, wf, wb, armed, version)
        self.tables = {}            # kind -> (device table, n, total rows, keys)
        self.dirty = True
        self.epoch = 0              # bumped by weight_bank_invalidate(): pairs armed in an earlier epoch are never served
        self.pinned = []            # buffers a stream capture saw (a replayed graph reads and writes them by raw pointer): never freed

    @staticmethod
    def _kind_of(p):
        if p.dim() == 5 and tuple(p.shape[2:]) == (3, 3, 3):
            return 'conv'
        if p.dim() in (3, 4) and all(k == 1 for k in p.shape[2:]):
            return 'pw'
        return None

    def register(self, model):
        import torch.nn as nn
        for m in model.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)) and m.weight is not None and self._kind_of(m.weight) and id(m.weight) not in self.seen:
                self.seen.add(id(m.weight))
                self.params.append(self._weakref.ref(m.weight))
                self.dirty = True

    def take(self, kind, w, nsplit=2):
        key = (kind, w.data_ptr(), (int(w.shape[0]), int(w.shape[1])), int(nsplit))
        e = self.entries.get(key)
        if e is not None and e['armed']:
            e['armed'] = False
            p = e['param']()
            if p is not None and p._version == e['version'] and p.data_ptr() == key[1] and e['epoch'] == self.epoch:
                if e['wf'].is_cuda and torch.cuda.is_current_stream_capturing():
                    self._pin(e)
                return e['wf'], e['wb']
        if key not in self.wanted:
            if len(self.wanted) >= 4096:                # (keys of temporaries -- a weight that is copied to be made contiguous has a new
                self.wanted.clear()                     # address per call -- must not pile up; registered weights are re-noted at once)
            self.wanted.add(key)
            self.dirty = True
        return None

    def _pin(self, e):
        if not e.get('pinned'):
            e['pinned'] = True
            self.pinned.append((e['wf'], e['wb']))

    def _rebuild(self):
        lib = self.be.lib
        # (kind, nsplit): nsplit 2 = the f16x2 pairs, 1 = the plain-bf16 pairs of the autocast mode
        live, by_kind = {}, {('conv', 2): [], ('pw', 2): [], ('conv', 1): [], ('pw', 1): []}
        for ref in self.params:
            p = ref()
            if p is None or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                continue
            kind = self._kind_of(p)
            for nsplit in (2, 1):
                key = (kind, p.data_ptr(), (int(p.shape[0]), int(p.shape[1])), nsplit)
                if key not in self.wanted:
                    continue
                e = self.entries.get(key)
                if e is None or e['param']() is not p:
                    co, ci = key[2]
                    nbytes = lib.pvcnn_conv3d_weight_split_bytes if kind == 'conv' else lib.pvcnn_pwconv_weight_split_bytes
                    e = {'param': ref, 'armed': False, 'version': -1, 'epoch': -1,
                         'wf': torch.empty((nbytes(co, ci, 0, nsplit),), dtype=torch.uint8, device=p.device),
                         'wb': torch.empty((nbytes(co, ci, 1, nsplit),), dtype=torch.uint8, device=p.device)}
                live[key] = e
                by_kind[(kind, nsplit)].append((key, p, e))
        self.entries, self.tables = live, {}
        for (kind, nsplit), items in by_kind.items():
            if not items:
                continue
            fill = {('conv', 2): lib.pvcnn_conv3d_weight_split_pair_entry, ('pw', 2): lib.pvcnn_pwconv_weight_split_pair_entry,
                    ('conv', 1): lib.pvcnn_conv3d_weight_split_pair_entry_bf16, ('pw', 1): lib.pvcnn_pwconv_weight_split_pair_entry_bf16}[(kind, nsplit)]
            host = torch.zeros((len(items), 10), dtype=torch.int64)
            rows = 0
            for i, (key, p, e) in enumerate(items):
                n = fill(_p(p), key[2][0], key[2][1], _p(e['wf']), _p(e['wb']), ctypes.c_void_p(host[i].data_ptr()))
                if n < 0:
                    raise RuntimeError('weight bank: bad entry')
                host[i, 9] = rows
                rows += n
            dev = items[0][1].device
            self.tables[(kind, nsplit)] = (host.to(dev), len(items), rows, [k for k, _, _ in items], dev)
        self.dirty = False

    def refresh(self):
        if self.dirty:
            if torch.cuda.is_current_stream_capturing():        # no allocation / host-to-device copy inside a capture: this step's layers
                for e in self.entries.values():                 # split their own weights (take() finds nothing armed)
                    e['armed'] = False
                return
            self._rebuild()
        capturing = torch.cuda.is_current_stream_capturing()
        lib = self.be.lib
        launches = {('conv', 2): lib.pvcnn_conv3d_weight_split_pair_batch, ('pw', 2): lib.pvcnn_pwconv_weight_split_pair_batch,
                    ('conv', 1): lib.pvcnn_conv3d_weight_split_pair_batch_bf16, ('pw', 1): lib.pvcnn_pwconv_weight_split_pair_batch_bf16}
        for kind, (table, n, rows, keys, dev) in self.tables.items():
            launch = launches[kind]
            with _Launch(table) as s:
                _lib.check(launch(_p(table), n, rows, s), 'weight_split_pair_batch')
            if capturing:                               # the captured launch walks this table and writes every entry's buffers on replay
                self.pinned.append(table)
            for key in keys:
                e = self.entries[key]
                if capturing:
                    self._pin(e)
                p = e['param']()
                if p is not None and p.data_ptr() == key[1]:
                    e['armed'], e['version'], e['epoch'] = True, p._version, self.epoch
                else:                                   # the parameter moved or died: rebuild before the next refresh
                    e['armed'] = False
                    self.dirty = True


_backend = HipBackend()
