#!/usr/bin/env python3
"""Is a matrix-core kernel limited by the power envelope or by its schedule?  Same binary, three input fills: random normal, the constant
1.0 (every operand bit pattern equal: the matrix pipes toggle as little as they can while doing all the work) and 0/1 integers.  A kernel
that runs much faster on the constant fill is at the power-limited MFMA rate (profiles/ab/r03v_mfma_power_limit.md).  (Round 3 used
zeros for the quiet fill; since round 4 the f16x2 Conv3d kernels skip all-zero tiles, so zeros no longer execute the work.)
usage: mfma_fill_probe.py [--kinds randn,const,small-int]     -- one kind per process when run under rocprofv3 --pmc, so that the
per-kernel counter averages (GRBM_GUI_ACTIVE / 8 / duration = effective clock) belong to ONE fill."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend(); dev = 'cuda:0'
KINDS = (sys.argv[sys.argv.index('--kinds') + 1].split(',') if '--kinds' in sys.argv else ['randn', 'const', 'small-int'])


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fill(kind, *shape, scale=1.0):
    if kind == 'randn':
        return torch.randn(*shape, device=dev) * scale
    if kind == 'const':
        return torch.ones(*shape, device=dev) * scale
    return torch.randint(0, 2, shape, device=dev).float()


for (b, ci, co, r) in [(16, 64, 64, 16), (16, 128, 128, 16), (16, 64, 64, 32)]:
    for kind in KINDS:
        x, w, gy = fill(kind, b, ci, r, r, r), fill(kind, co, ci, 3, 3, 3, scale=0.1), fill(kind, b, co, r, r, r)
        ax, ag = be.conv_amax(x), be.conv_amax(gy)
        wf = be._conv_wsplit(w, False, 2)
        f = t(lambda: be.conv3d_igemm_split(x, wf, None, co, 2, False, ax))
        g = t(lambda: be.conv3d_backward_weight_f16(x, gy, ax, ag))
        print('conv3d', (b, ci, co, r), kind, 'fwd %.1f us' % f, 'bwd-weight %.1f us' % g, flush=True)
for (b, ci, co, n) in [(16, 1472, 512, 4096), (16, 128, 1024, 4096)]:
    for kind in KINDS:
        x, w, gy = fill(kind, b, ci, n), fill(kind, co, ci, scale=0.1), fill(kind, b, co, n)
        ax, ag = be.pw_amax(x), be.pw_amax(gy)
        wf = be._pw_wsplit(w, False, 2)
        f = t(lambda: be.pwconv_gemm_split(x, wf, None, co, 2, False, ax))
        g = t(lambda: be.pwconv_backward_weight_f16(x, gy, ax, ag))
        print('1x1', (b, ci, co, n), kind, 'fwd %.1f us' % f, 'bwd-weight %.1f us' % g, flush=True)
