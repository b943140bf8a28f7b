#!/usr/bin/env python3
"""One-process check + timing of the EXPERIMENTAL pipelined gather (run with PVCNN_GATHER_PIPE=1): the pipelined kernel (taken for
16-byte aligned inputs) against the classic kernel (forced by a misaligned coordinate view: VEC = 1 path), bit for bit, then its
device time at the bench's roofline shape."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pvcnn_amd.modules.functional.backend import HipBackend

be = HipBackend()
dev = 'cuda:0'
torch.manual_seed(0)
b, c, n, r = 16, 64, 4096, 32
coords = torch.rand(b, 3, n, device=dev) * (r - 1)
feat = torch.randn(b, c, r ** 3, device=dev)
gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
mean, rstd = torch.randn(c, device=dev) * 0.2, torch.rand(c, device=dev) + 0.5
addend = torch.randn(b, c, n, device=dev)
# a copy of the coordinates that starts 4 bytes into its storage: not 16-byte aligned -> the classic VEC = 1 kernel
store = torch.empty(coords.numel() + 1, device=dev)
mis = store[1:].view_as(coords)
mis.copy_(coords)
a = be.trilinear_devoxelize_bnact_forward(r, True, coords, feat, gamma, beta, mean, rstd, 0.1, addend)
ref = be.trilinear_devoxelize_bnact_forward(r, True, mis, feat, gamma, beta, mean, rstd, 0.1, addend)
a2 = be.trilinear_devoxelize_forward(r, True, coords, feat)
ref2 = be.trilinear_devoxelize_forward(r, True, mis, feat)
same = all(torch.equal(x, y) for x, y in zip(a + a2, ref + ref2))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(5):
    be.trilinear_devoxelize_bnact_forward(r, False, coords, feat, gamma, beta, mean, rstd, 0.1, addend)
torch.cuda.synchronize()
ev[0].record()
for _ in range(50):
    be.trilinear_devoxelize_bnact_forward(r, False, coords, feat, gamma, beta, mean, rstd, 0.1, addend)
ev[1].record()
torch.cuda.synchronize()
print(json.dumps({'PVCNN_GATHER_PIPE': os.environ.get('PVCNN_GATHER_PIPE'), 'bit_identical_to_classic_kernel': bool(same),
                  'devox_bnact_fwd_eval_us': round(ev[0].elapsed_time(ev[1]) * 1e3 / 50, 1), 'shape_BCNR': [b, c, n, r]}), flush=True)
