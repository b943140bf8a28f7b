#!/usr/bin/env python3
"""Does the headline gather kernel slow down when its input grid was just written (dirty in L2/MALL)?
Run under `rocprofv3 --kernel-trace --stats`; compare the gather kernel's average in the three phases."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
dev = 'cuda:0'
B, C, N, R = 16, 64, 4096, 32
torch.manual_seed(0)
grid = torch.randn(B, C, R ** 3, device=dev)
coords = torch.rand(B, 3, N, device=dev) * (R - 1)
mode = sys.argv[1] if len(sys.argv) > 1 else 'alone'
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(40):
    if mode == 'writer':
        g = grid * 1.0001                     # fresh 134 MB tensor written right before the gather
    elif mode == 'mfma':
        a = torch.randn(4096, 4096, device=dev); (a @ a).sum()   # MFMA-heavy kernel first, same grid
        g = grid
    else:
        g = grid
    be.trilinear_devoxelize_forward(R, True, coords, g)
torch.cuda.synchronize()
print('done', mode)
