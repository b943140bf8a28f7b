#!/usr/bin/env python3
"""PVCNN_GATHER_PIPE=1: the pipelined gather vs the classic kernel (forced by misaligned coordinates) over ragged shapes, one process."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend(); dev = 'cuda:0'; r = 32
torch.manual_seed(1)
res = []
for b, c, n in [(2, 5, 1024), (1, 1, 4092), (8, 71, 4096), (3, 130, 2048), (64, 64, 2048), (16, 9, 4096), (2, 3, 4)]:
    coords = torch.rand(b, 3, n, device=dev) * (r - 1)
    coords[:, :, : max(1, n // 8)] = torch.round(coords[:, :, : max(1, n // 8)])
    feat = torch.randn(b, c, r ** 3, device=dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    mean, rstd = torch.randn(c, device=dev) * 0.2, torch.rand(c, device=dev) + 0.5
    addend = torch.randn(b, c, n, device=dev)
    store = torch.empty(coords.numel() + 1, device=dev)
    mis = store[1:].view_as(coords); mis.copy_(coords)
    ok = True
    for train in (True, False):
        a = be.trilinear_devoxelize_bnact_forward(r, train, coords, feat, gamma, beta, mean, rstd, 0.1, addend)
        ref = be.trilinear_devoxelize_bnact_forward(r, train, mis, feat, gamma, beta, mean, rstd, 0.1, addend)
        a2 = be.trilinear_devoxelize_forward(r, train, coords, feat)
        ref2 = be.trilinear_devoxelize_forward(r, train, mis, feat)
        a3 = be.trilinear_devoxelize_bnact_forward(r, train, coords, feat, None, None, mean, rstd, 0.0, None)
        ref3 = be.trilinear_devoxelize_bnact_forward(r, train, mis, feat, None, None, mean, rstd, 0.0, None)
        ok = ok and all(torch.equal(x, y) for x, y in zip(a + a2 + a3, ref + ref2 + ref3))
    res.append([b, c, n, bool(ok)])
print(json.dumps({'PVCNN_GATHER_PIPE': os.environ.get('PVCNN_GATHER_PIPE'), 'cases_BCN_ok': res, 'all_ok': all(x[3] for x in res)}), flush=True)
