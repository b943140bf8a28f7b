#!/usr/bin/env python3
"""The f16x2 Conv3d launch alone, forward and backward-data, N times at one shape; prints the per-launch time (HIP events)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
b, ci, co, r = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '16x64x64x32').split('x'))
x = torch.randn(b, ci, r, r, r, device='cuda:0')
if '--relu' in sys.argv:
    x = torch.nn.functional.leaky_relu(x, 0.1)
gy = torch.randn(b, co, r, r, r, device='cuda:0')
w = torch.randn(co, ci, 3, 3, 3, device='cuda:0') * 0.05
bias = torch.randn(co, device='cuda:0')
wf, wb = be._conv_wsplit(w, False, 2), be._conv_wsplit(w, True, 2)
ax, ag = be.conv_amax(x), be.conv_amax(gy)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


f = t(lambda: be.conv3d_igemm_split(x, wf, bias, co, 2, True, ax))
d = t(lambda: be.conv3d_igemm_split(gy, wb, None, ci, 2, False, ag))
print(json.dumps({"BCiCoR": [b, ci, co, r], "wide": os.environ.get("PVCNN_CONV_WIDE", "1") + "/" + os.environ.get("PVCNN_CONV_WIDE16", "1"), 'ablate': os.environ.get('PVCNN_CONV_ABLATE', '0'),
                  'fwd_us': round(f, 1), 'bwd_data_us': round(d, 1)}))
