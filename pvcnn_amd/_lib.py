"""ctypes binding of libpvcnn_hip.so (C ABI: include/pvcnn_hip.h).

The library is built in-tree (pvcnn_amd/csrc/libpvcnn_hip.so) by `__graft_entry__.build()` or
`make -C pvcnn_amd/csrc`.  There is NO fallback: if it is missing or an entry point fails, the
caller gets an exception -- the product path never degrades to a CPU / eager implementation.

`import torch` happens before the dlopen on purpose: torch has already mapped its own
libamdhip64.so.7, and the dynamic loader resolves this library's DT_NEEDED entry of the same
soname to that copy, so kernels, streams and device pointers share ONE HIP runtime.
"""
import ctypes
import functools
import os

import torch  # noqa: F401  (must be imported before the dlopen, see above)

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(_CSRC, 'libpvcnn_hip.so')
ABI_VERSION = 12

_vp, _i, _f, _sz, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long

# name -> (restype, argtypes); mirrors include/pvcnn_hip.h one to one
SIGNATURES = {
    'pvcnn_version': (_i, []),
    'pvcnn_last_error_string': (ctypes.c_char_p, []),
    'pvcnn_voxel_coords': (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    'pvcnn_voxel_coords_tail': (_i, [_vp, ctypes.c_long, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    'pvcnn_avg_voxelize_fwd_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_avg_voxelize_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_avg_voxelize_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_trilinear_devox_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'pvcnn_trilinear_devox_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_trilinear_devox_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'pvcnn_avg_voxelize_plan_bytes': (_sz, [_i, _i, _i]),
    'pvcnn_avg_voxelize_plan_scratch_bytes': (_sz, [_i, _i, _i]),
    'pvcnn_avg_voxelize_plan': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    'pvcnn_avg_voxelize_apply': (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_pvconv_plans_scratch_bytes': (_sz, [_i, _i, _i]),
    'pvcnn_pvconv_plans': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    'pvcnn_trilinear_devox_bwd_plan_bytes': (_sz, [_i, _i, _i]),
    'pvcnn_trilinear_devox_bwd_plan_scratch_bytes': (_sz, [_i, _i, _i]),
    'pvcnn_trilinear_devox_bwd_plan': (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    'pvcnn_trilinear_devox_bwd_apply': (_i, [_vp, ctypes.c_long, _vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_ball_query': (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    'pvcnn_grouping_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_grouping_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'pvcnn_grouping_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'pvcnn_gather_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_gather_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_gather_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'pvcnn_fps': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'pvcnn_mask_select': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'pvcnn_three_nn_interp_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'pvcnn_three_nn_interp_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_three_nn_interp_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'pvcnn_conv3d_weight_transform': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'pvcnn_conv3d_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_conv3d_bwd_weight_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_conv3d_bwd_weight': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_conv3d_weight_split_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_conv3d_weight_split': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_conv3d_weight_split_pair': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'pvcnn_conv3d_fwd_split_stats_parts': (_sz, [_i, _i, _i, _i]),
    'pvcnn_conv3d_fwd_split_route': (_i, [_i, _i, _i, _i, _i]),
    'pvcnn_absmax_bits': (_i, [_vp, _sz, _vp, _vp]),
    'pvcnn_absmax_tiles_count': (_sz, [_i, ctypes.c_long, _i]),
    'pvcnn_absmax_tiles': (_i, [_vp, _i, _i, ctypes.c_long, _i, _vp, _vp, _vp]),
    'pvcnn_conv3d_bwd_weight_f16_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_conv3d_bwd_weight_f16': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_conv3d_fwd_split': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'pvcnn_pwconv_transpose': (_i, [_vp, _i, _i, _vp, _vp]),
    'pvcnn_pwconv_fwd': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_pwconv_weight_split_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_pwconv_weight_split': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'pvcnn_pwconv_weight_split_pair': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'pvcnn_pwconv_fwd_split_stats_parts': (_sz, [_i, _i]),
    'pvcnn_pwconv_fwd_split': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'pvcnn_pwconv_bwd_weight_f16_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_pwconv_bwd_weight_f16': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_pwconv_bwd_weight_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pvcnn_pwconv_bwd_weight': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_bnact_workspace_bytes': (_sz, [_i, _i, _i]),
    'pvcnn_bnact_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp, _f, _vp]),
    'pvcnn_dropout_keep_mask': (_i, [_vp, _f, _l, _vp, _vp]),
    'pvcnn_conv3d_fwd_stats_parts': (_sz, [_i, _i, _i]),
    'pvcnn_conv3d_fwd_stats': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'pvcnn_pwconv_fwd_stats_parts': (_sz, [_i, _i]),
    'pvcnn_pwconv_fwd_stats': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'pvcnn_bn_finalize': (_i, [_vp, _i, ctypes.c_long, ctypes.c_double, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _vp, _vp]),
    'pvcnn_bn_stats': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_trilinear_devox_bnact_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pvcnn_bnact_slices': (_i, [_i]),
    'pvcnn_conv3d_weight_split_pair_entry': (_l, [_vp, _i, _i, _vp, _vp, _vp]),
    'pvcnn_conv3d_weight_split_pair_batch': (_i, [_vp, _i, _l, _vp]),
    'pvcnn_pwconv_weight_split_pair_entry': (_l, [_vp, _i, _i, _vp, _vp, _vp]),
    'pvcnn_pwconv_weight_split_pair_batch': (_i, [_vp, _i, _l, _vp]),
    'pvcnn_conv3d_weight_split_pair_entry_bf16': (_l, [_vp, _i, _i, _vp, _vp, _vp]),
    'pvcnn_conv3d_weight_split_pair_batch_bf16': (_i, [_vp, _i, _l, _vp]),
    'pvcnn_pwconv_weight_split_pair_entry_bf16': (_l, [_vp, _i, _i, _vp, _vp, _vp]),
    'pvcnn_pwconv_weight_split_pair_batch_bf16': (_i, [_vp, _i, _l, _vp]),
    'pvcnn_neighbor_max_supported': (_i, [_i]),
    'pvcnn_neighbor_max_fwd': (_i, [_vp, _l, _i, _vp, _vp, _vp]),
    'pvcnn_neighbor_max_bwd': (_i, [_vp, _vp, _l, _i, _vp, _vp]),
    'pvcnn_row_argmax': (_i, [_vp, _l, _i, _vp, _vp, _vp]),
    'pvcnn_bnact_apply_rowmax': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _l, _vp, _i, _vp, _vp]),
    'pvcnn_row_keys_decode': (_i, [_vp, _vp, _l, _i, _vp, _vp, _i, _l, _vp]),
    'pvcnn_frustum_box_loss_grad_floats': (_sz, [_i, _i, _i]),
    'pvcnn_frustum_box_loss': (_i, [_vp] * 15 + [_i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp]),
    'pvcnn_se_excite_fwd': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pvcnn_se_excite_bwd': (_i, [_vp, _i] + [_vp] * 9 + [_i, _i, _i, _f] + [_vp] * 7),
    'pvcnn_bnact_partial_sums': (_i, [_vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    'pvcnn_bnact_bwd_apply': (_i, [_vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp]),
    'pvcnn_bnact_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pvcnn_bnact_bwd_strided': (_i, [_vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _f, _vp, _vp]),
    'pvcnn_concat_points': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'pvcnn_dense_bn_relu_supported': (_i, [_i, _i, _i]),
    'pvcnn_dense_bn_relu_fwd': (_i, [_vp] * 8 + [_i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'pvcnn_dense_bn_relu_bwd': (_i, [_vp] * 7 + [_i, _i, _i] + [_vp] * 6),
    'pvcnn_adam_step': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _i, _vp]),
    'pvcnn_trilinear_devox_bwd_strided': (_i, [_vp, ctypes.c_long, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
}

_lib = None


class PvcnnHipError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library (once) and attach the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PvcnnHipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C {_CSRC}`.  There is no CPU fallback for the PVConv hot path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the ABI lost a symbol
        fn.restype, fn.argtypes = res, args
    got = lib.pvcnn_version()
    if got != ABI_VERSION:
        raise PvcnnHipError(f'libpvcnn_hip.so ABI version {got}, binding expects {ABI_VERSION}: rebuild')
    # the `*_bytes` / `*_parts` queries are pure functions of a few ints and are asked before every launch: memoise them
    # (one dict lookup instead of a foreign call on the eager step's critical path)
    for name, (res, _args) in SIGNATURES.items():
        if res is _sz:
            setattr(lib, name, functools.lru_cache(maxsize=4096)(getattr(lib, name)))
    _lib = lib
    return lib


def sources_digest():
    """sha1 over the kernel sources the library is built from (csrc/*.hip, csrc/*.h, include/pvcnn_hip.h), 12 hex digits.  Evidence
    files that describe kernels (profiles/kernel_durations*.json, pmc_traffic.json) carry the digest they were taken at; bench.py only
    prices on them while it equals the digest of the sources it runs (the GPU box has no .git to ask for a commit)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    names = sorted(glob.glob(os.path.join(_CSRC, '*.hip')) + glob.glob(os.path.join(_CSRC, '*.h')))
    names.append(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'pvcnn_hip.h'))
    for name in names:
        h.update(os.path.basename(name).encode())
        with open(name, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def check(rc, what):
    """Turn a non-zero ABI return code into an exception carrying the library's message."""
    if rc != 0:
        msg = load().pvcnn_last_error_string().decode('utf-8', 'replace')
        raise PvcnnHipError(f'{what} failed (code {rc}): {msg}')
