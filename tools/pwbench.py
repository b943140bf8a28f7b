#!/usr/bin/env python3
"""Speed of the 1x1-convolution GEMM kernels (csrc/pointwise.hip) at PVCNN's SharedMLP shapes, next to torch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from pvcnn_amd.modules.functional.backend import HipBackend

be = HipBackend()
dev = 'cuda:0'


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
    return e0.elapsed_time(e1) / n


shapes = [(16, 9, 64, 4096), (16, 64, 64, 4096), (16, 64, 128, 4096), (16, 128, 1024, 4096), (16, 1472, 512, 4096), (16, 512, 256, 4096)]
if '--shapes' in sys.argv:
    shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[sys.argv.index('--shapes') + 1].split(',')]
for (b, ci, co, n) in shapes:
    x = torch.randn(b, ci, n, device=dev)
    w = torch.randn(co, ci, device=dev) * 0.1
    bias = torch.randn(co, device=dev)
    gy = torch.randn(b, co, n, device=dev)
    fl = 2.0 * b * n * ci * co
    ax, ag = be.pw_amax(x), be.pw_amax(gy)
    wf, wb = be._pw_wsplit(w, False, 2), be._pw_wsplit(w, True, 2)
    f2 = t(lambda: be.pwconv_gemm_split(x, wf, bias, co, 2, False, ax))        # the GEMM launch alone
    ax1 = be.absmax_bits(x)
    f2s = t(lambda: be.pwconv_gemm_split(x, wf, bias, co, 2, False, ax1))      # ... with ONE scale for the tensor (amax_seg = 0)
    tam = t(lambda: be.pw_amax(x))
    d2 = t(lambda: be.pwconv_gemm_split(gy, wb, None, ci, 2, False, ag))
    f2all = t(lambda: be.pwconv_forward_split(x, w, bias, 2))                  # + weight split + absmax
    e2 = ((be.pwconv_forward_split(x, w, bias, 2).double() - F.conv1d(x.double(), w.double().view(co, ci, 1), bias.double())).abs().max()
          / F.conv1d(x.double(), w.double().view(co, ci, 1), bias.double()).abs().max()).item()
    g2 = t(lambda: be.pwconv_backward_weight_f16(x, gy, ax, ag, with_bias=True))
    gwr = torch.einsum('bon,bcn->oc', gy.double(), x.double())
    eg = ((be.pwconv_backward_weight_f16(x, gy, ax, ag).double() - gwr).abs().max() / gwr.abs().max()).item()
    print(json.dumps({'f16x2_wgrad_BCiCoN': [b, ci, co, n], 'bwd_w_ms': round(g2, 4), 'eff_TF': round(fl / g2 / 1e9, 1), 'err': eg}), flush=True)
    print(json.dumps({'f16x2_BCiCoN': [b, ci, co, n], 'fwd_ms': round(f2, 4), 'fwd_eff_TF': round(fl / f2 / 1e9, 1), 'bwd_data_ms': round(d2, 4),
                      'fwd_with_split_and_absmax_ms': round(f2all, 4), 'fwd_single_scale_ms': round(f2s, 4), 'amax_tiles_ms': round(tam, 4), 'err_f16x2': e2}), flush=True)
    f3 = t(lambda: be.pwconv_forward_split(x, w, bias, 3))
    d3 = t(lambda: be.pwconv_backward_data_split(gy, w, 3))
    f1 = t(lambda: be.pwconv_forward_split(x, w, bias, 1))
    ref = F.conv1d(x.double(), w.double().view(co, ci, 1), bias.double())
    e3 = ((be.pwconv_forward_split(x, w, bias, 3).double() - ref).abs().max() / ref.abs().max()).item()
    e1 = ((be.pwconv_forward_split(x, w, bias, 1).double() - ref).abs().max() / ref.abs().max()).item()
    e0 = ((be.pwconv_forward(x, w, bias).double() - ref).abs().max() / ref.abs().max()).item()
    gxr = torch.einsum('oc,bon->bcn', w.double(), gy.double())
    ed3 = ((be.pwconv_backward_data_split(gy, w, 3).double() - gxr).abs().max() / gxr.abs().max()).item()
    print(json.dumps({'split_BCiCoN': [b, ci, co, n], 'bf16x3_fwd_ms': round(f3, 4), 'bf16x3_fwd_eff_TF': round(fl / f3 / 1e9, 1), 'bf16x3_bwd_data_ms': round(d3, 4),
                      'bf16_fwd_ms': round(f1, 4), 'err_fp32mfma': e0, 'err_bf16x3': e3, 'err_bf16x3_bwd_data': ed3, 'err_bf16': e1}), flush=True)
    f = t(lambda: be.pwconv_forward(x, w, bias))
    d = t(lambda: be.pwconv_backward_data(gy, w))
    g = t(lambda: be.pwconv_backward_weight(x, gy, with_bias=True))
    w3 = w.view(co, ci, 1).clone().requires_grad_()
    x3 = x.clone().requires_grad_()
    tf = t(lambda: F.conv1d(x, w3, bias))
    y = F.conv1d(x3, w3, bias)
    tb = t(lambda: torch.autograd.grad(y, (x3, w3), gy, retain_graph=True))
    print(json.dumps({'BCiCoN': [b, ci, co, n], 'GF': round(fl / 1e9, 1), 'fwd_ms': round(f, 4), 'fwd_TF': round(fl / f / 1e9, 1),
                      'bwd_data_ms': round(d, 4), 'bwd_data_TF': round(fl / d / 1e9, 1), 'bwd_w_ms': round(g, 4),
                      'bwd_w_TF': round(fl / g / 1e9, 1), 'torch_fwd_ms': round(tf, 4), 'torch_bwd_both_ms': round(tb, 4)}), flush=True)
