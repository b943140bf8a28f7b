#!/usr/bin/env python3
"""One split-conv launch shape in a loop (for rocprofv3 --pmc passes): python tools/convprobe.py B Ci Co R nsplit"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pvcnn_amd import _lib
lib = _lib.load(); dev = 'cuda:0'
b, ci, co, r, ns = [int(v) for v in sys.argv[1:6]]
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(b, ci, r, r, r, device=dev); w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.1; bias = torch.randn(co, device=dev)
y = torch.empty(b, co, r, r, r, device=dev)
nb = lib.pvcnn_conv3d_weight_split_bytes(co, ci, 0, ns); wts = torch.empty(nb, dtype=torch.uint8, device=dev)
lib.pvcnn_conv3d_weight_split(P(w), co, ci, 0, ns, P(wts), S())
for _ in range(5):
    lib.pvcnn_conv3d_fwd_split(P(x), P(wts), P(bias), b, ci, co, r, ns, P(y), None, S())
torch.cuda.synchronize()
