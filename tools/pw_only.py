#!/usr/bin/env python3
"""The f16x2 1x1 GEMM launch alone, forward and backward-data, N times at one shape (for counter passes: tools/calls_r06)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '16x1472x512x4096').split('x'))
relu = '--relu' in sys.argv           # operands like the step's: a BatchNorm + ReLU output (half zeros) instead of a normal fill
b, ci, co, n = shape
x = torch.randn(b, ci, n, device='cuda:0')
gy = torch.randn(b, co, n, device='cuda:0')
if relu:
    x = torch.relu(x)
w = torch.randn(co, ci, device='cuda:0') * 0.1
bias = torch.randn(co, device='cuda:0')
wf, wb = be._pw_wsplit(w, False, 2), be._pw_wsplit(w, True, 2)
ax, ag = be.pw_amax(x), be.pw_amax(gy)
for _ in range(12):
    be.pwconv_gemm_split(x, wf, bias, co, 2, False, ax)
    be.pwconv_gemm_split(gy, wb, None, ci, 2, False, ag)
torch.cuda.synchronize()
