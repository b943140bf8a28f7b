import sys, torch
sys.path.insert(0, '.')
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend(); dev = 'cuda:0'
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (b, ci, co, r) in [(16, 64, 64, 16), (16, 128, 128, 16), (16, 64, 64, 32)]:
    for kind in ('randn', 'zeros', 'small-int'):
        if kind == 'randn': x = torch.randn(b, ci, r, r, r, device=dev); w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.1
        elif kind == 'zeros': x = torch.zeros(b, ci, r, r, r, device=dev); w = torch.zeros(co, ci, 3, 3, 3, device=dev)
        else: x = torch.randint(0, 2, (b, ci, r, r, r), device=dev).float(); w = torch.randint(0, 2, (co, ci, 3, 3, 3), device=dev).float()
        ax = be.conv_amax(x)
        wf = be._conv_wsplit(w, False, 2)
        us = t(lambda: be.conv3d_igemm_split(x, wf, None, co, 2, False, ax))
        print(b, ci, co, r, kind, round(us, 1), 'us', flush=True)
