#!/usr/bin/env python3
"""Steady-state time of the voxel_layers convolutions (PVCNN 1xC, B=16) through torch/MIOpen."""
import json, sys, time, torch
import torch.nn.functional as F
bench = '--find' in sys.argv
torch.backends.cudnn.benchmark = bench
dev = 'cuda:0'
shapes = [(16, 9, 64, 32), (16, 64, 64, 32), (16, 64, 64, 16), (16, 64, 128, 16), (16, 128, 128, 16)]
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
for (b, ci, co, r) in shapes:
    x = torch.randn(b, ci, r, r, r, device=dev, requires_grad=True)
    w = torch.randn(co, ci, 3, 3, 3, device=dev, requires_grad=True)
    y = F.conv3d(x, w, padding=1); g = torch.randn_like(y)
    fl = 2 * b * r ** 3 * 27 * ci * co
    t0 = time.time()
    fwd = t(lambda: F.conv3d(x, w, padding=1))
    bwd_d = t(lambda: torch.autograd.grad(F.conv3d(x, w.detach(), padding=1), x, g)) - fwd
    bwd_w = t(lambda: torch.autograd.grad(F.conv3d(x.detach(), w, padding=1), w, g)) - fwd
    xl = x.detach().contiguous(memory_format=torch.channels_last_3d); wl = w.detach().contiguous(memory_format=torch.channels_last_3d)
    fwd_cl = t(lambda: F.conv3d(xl, wl, padding=1))
    print(json.dumps({'BCiCoR': [b, ci, co, r], 'find': bench, 'GF_fwd': round(fl / 1e9, 1), 'fwd_ms': round(fwd, 3), 'fwd_TF': round(fl / fwd / 1e9, 1),
                      'bwd_data_ms': round(bwd_d, 3), 'bwd_data_TF': round(fl / max(bwd_d, 1e-3) / 1e9, 1), 'bwd_w_ms': round(bwd_w, 3),
                      'bwd_w_TF': round(fl / max(bwd_w, 1e-3) / 1e9, 1), 'fwd_channels_last_ms': round(fwd_cl, 3), 'wall_s': round(time.time() - t0, 1)}), flush=True)
# BatchNorm3d + LeakyReLU passes on the largest grid
x = torch.randn(16, 64, 32, 32, 32, device=dev, requires_grad=True)
bn = torch.nn.BatchNorm3d(64, eps=1e-4).to(dev).train(); act = torch.nn.LeakyReLU(0.1, True)
g = torch.randn_like(x)
f = t(lambda: act(bn(x)))
fb = t(lambda: torch.autograd.grad(act(bn(x)), x, g))
print(json.dumps({'bn_lrelu_16x64x32^3': {'fwd_ms': round(f, 3), 'fwd_bwd_ms': round(fb, 3), 'tensor_MB': 134.2}}))
