"""The C-ABI library loads and exports exactly what include/pvcnn_hip.h declares (no GPU needed:
no compute call is made here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'pvcnn_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'PVCNN_API\s+[\w\s\*]+?\b(pvcnn_\w+)\s*\(', text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    # one entry point per reference native function (bindings.cpp:10-37) + version/error/workspace
    for name in ['pvcnn_avg_voxelize_fwd', 'pvcnn_avg_voxelize_bwd', 'pvcnn_trilinear_devox_fwd',
                 'pvcnn_trilinear_devox_bwd', 'pvcnn_ball_query', 'pvcnn_grouping_fwd', 'pvcnn_grouping_bwd',
                 'pvcnn_gather_fwd', 'pvcnn_gather_bwd', 'pvcnn_fps', 'pvcnn_three_nn_interp_fwd',
                 'pvcnn_three_nn_interp_bwd', 'pvcnn_version', 'pvcnn_last_error_string',
                 'pvcnn_avg_voxelize_fwd_workspace_bytes']:
        assert name in syms


def test_library_exports_every_declared_symbol():
    from pvcnn_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'libpvcnn_hip.so not built (run __graft_entry__.build())'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in pvcnn_hip.h but not exported'


def test_binding_table_matches_header():
    from pvcnn_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.pvcnn_version() == _lib.ABI_VERSION
    assert lib.pvcnn_last_error_string() is not None
    # one-shot workspace = the plan (prefix offsets, sorted entries, count-sorted lane order) + the sort's scratch
    plan = lib.pvcnn_avg_voxelize_plan_bytes(16, 4096, 32)
    assert plan >= 16 * 32768 * 4 + 16 * 4096 * 8 + 16 * 32768 * 2
    assert lib.pvcnn_avg_voxelize_fwd_workspace_bytes(16, 64, 4096, 32) >= plan + lib.pvcnn_avg_voxelize_plan_scratch_bytes(16, 4096, 32) - 16
    assert lib.pvcnn_avg_voxelize_plan_bytes(1, 4096, 128) == 0          # beyond 2^20 targets: no plan, one-shot atomic fallback


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the product backend raises instead of computing on the host."""
    import torch
    from pvcnn_amd.modules.functional.backend import HipBackend
    b = HipBackend()
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        b.avg_voxelize_forward(torch.rand(1, 2, 8), torch.zeros(1, 3, 8, dtype=torch.int32), 2)
    with pytest.raises(RuntimeError):
        b.trilinear_devoxelize_forward(2, True, torch.rand(1, 3, 8), torch.rand(1, 2, 8))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under pvcnn_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'pvcnn_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f'{f} imports the oracle'
                assert 'libpvcnn_oracle' not in src, f'{f} references the oracle library'


def test_batched_weight_split_entries_are_filled_on_the_host():
    """pvcnn_*_weight_split_pair_entry (host-only helpers of the batched image refresh): workgroup counts, the wexp pointers behind the
    images, the packed tile heights -- and bad arguments are refused with -1 instead of a table entry."""
    import torch
    from pvcnn_amd import _lib
    lib = _lib.load()
    w, wf, wb = 0x100000, 0x200000, 0x300000                      # any non-null, 16-byte aligned addresses: nothing is dereferenced
    e = torch.zeros(10, dtype=torch.int64)
    vp = ctypes.c_void_p
    rows = lib.pvcnn_conv3d_weight_split_pair_entry(vp(w), 70, 33, vp(wf), vp(wb), vp(e.data_ptr()))
    assert rows == 128 + 64                                       # (padded) output rows of the forward + backward-data image
    img_f = lib.pvcnn_conv3d_weight_split_bytes(70, 33, 0, 2) - 128 * 4
    img_b = lib.pvcnn_conv3d_weight_split_bytes(70, 33, 1, 2) - 64 * 4
    assert e.tolist() == [w, wf, wf + img_f, wb, wb + img_b, 70, 33, 128, 0, 0]
    rows = lib.pvcnn_pwconv_weight_split_pair_entry(vp(w), 130, 70, vp(wf), vp(wb), vp(e.data_ptr()))
    assert rows == 256 + 128 and e[5:8].tolist() == [130, 70, 256] and e[8].item() == (128 | (128 << 32))
    assert e[2].item() == wf + lib.pvcnn_pwconv_weight_split_bytes(130, 70, 0, 2) - 256 * 4
    assert lib.pvcnn_conv3d_weight_split_pair_entry(vp(w), 0, 33, vp(wf), vp(wb), vp(e.data_ptr())) == -1
    assert lib.pvcnn_pwconv_weight_split_pair_entry(vp(w), 8, 8, vp(wf + 4), vp(wb), vp(e.data_ptr())) == -1     # misaligned image
    assert lib.pvcnn_conv3d_weight_split_pair_batch(None, 0, 0, None) == 0                                          # empty table: no launch
