#!/usr/bin/env python3
"""How much of a voxelised cloud's grid is empty, and what the zero-tile / zero-row shortcuts of the f16x2 Conv3d kernels buy on it.
(1) occupancy of the z rows of the first PVConv's grid for the bench's synthetic S3DIS batch; (2) forward and backward-weight launch
times on that grid with the row table (shortcuts live) and with a 1-word amax buffer (nothing skipped)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvcnn_amd import workload
from pvcnn_amd.modules import Voxelization
from pvcnn_amd.modules.functional.backend import _backend as be

dev = 'cuda:0'


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (b, n, r, ci, co) in [(16, 4096, 32, 9, 64), (16, 4096, 16, 64, 64), (16, 4096, 16, 64, 128), (8, 8192, 32, 9, 32), (8, 2048, 32, 6, 64)]:
    x, _ = workload.make_s3dis_batch(b, n, device=dev)
    feats = torch.randn(b, ci, n, device=dev)
    vox = Voxelization(r, normalize=True, eps=0).to(dev)
    grid, _ = vox(feats, x[:, :3, :])
    amax = be.conv_amax(grid)
    rows = amax[1:].view(b, r, r)
    occ = (rows != 0).float().mean().item()
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.1
    bias = torch.randn(co, device=dev)
    img = be._conv_wsplit(w, False, 2)
    one = amax[:1].clone()
    gy = torch.randn(b, co, r, r, r, device=dev)
    ga = be.conv_amax(gy)
    f_tab = t(lambda: be.conv3d_igemm_split(grid, img, bias, co, 2, True, amax))
    f_one = t(lambda: be.conv3d_igemm_split(grid, img, bias, co, 2, True, one))
    w_tab = t(lambda: be.conv3d_backward_weight_f16(grid, gy, amax, ga, with_bias=True))
    w_one = t(lambda: be.conv3d_backward_weight_f16(grid, gy, one, ga, with_bias=True))
    print(f'B={b} N={n} R={r} {ci}->{co}: non-empty z rows {occ:.3f}; forward {f_tab:.1f} us (row table) vs {f_one:.1f} us (no shortcut); '
          f'backward-weight {w_tab:.1f} vs {w_one:.1f} us', flush=True)
