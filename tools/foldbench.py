#!/usr/bin/env python3
"""A/B of the two opt-in fusions at kernel level (needs a GPU): what does applying BatchNorm3d + LeakyReLU inside the second
convolution cost in the consumers, and what does it save?  One JSON line per shape; device time of hipGraph-captured launches.

  plain  : bnact_apply (1 read + 1 write) + absmax (1 read) + conv3d fwd            | wgrad on the materialised activation
  folded : bnact_absmax (1 read)                        + conv3d fwd with XF staging | wgrad with XF staging
  amax   : bnact_backward with / without the gradient maximum riding on the apply pass (vs a separate absmax pass)

usage: python tools/foldbench.py [--shapes 16x64x64x32,16x64x64x16,16x128x128x16]
Round-2 reading (profiles/r02_fold_ab.md): folded forward 274 vs 179 us and wgrad 344 vs 220 us at 16x64x64x32 -- the fold loses.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pvcnn_amd.modules.functional.backend import HipBackend
from tools.convcheck import graph_time

dev = 'cuda:0'


def main():
    be = HipBackend()
    shapes = [(16, 64, 64, 32), (16, 64, 64, 16), (16, 128, 128, 16)]
    if '--shapes' in sys.argv:
        shapes = [tuple(int(v) for v in t.split('x')) for t in sys.argv[sys.argv.index('--shapes') + 1].split(',')]
    torch.manual_seed(0)
    for b, ci, co, r in shapes:
        x = torch.randn(b, ci, r, r, r, device=dev) * 2 + 0.3           # the first convolution's raw output
        gy = torch.randn(b, co, r, r, r, device=dev) * 1e-3
        w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
        bias = torch.randn(co, device=dev)
        gamma, beta = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.3
        mean, rstd = torch.randn(ci, device=dev) * 0.2 + 0.3, torch.rand(ci, device=dev) + 0.5
        bn = (gamma, beta, mean, rstd, 0.1)
        x3 = x.view(b, ci, -1)
        act = be.bnact_forward(x3, gamma, beta, None, None, False, 0.0, 0.0, 0.1, stats=(mean, rstd))[0].view(x.shape)
        wts = be._conv_wsplit(w, False, 2)
        am_act, am_x, am_gy = be.absmax_bits(act), be.bnact_absmax_bits(x, bn), be.absmax_bits(gy)
        us = lambda fn: round(graph_time(fn) * 1e3, 1)
        row = {'BCiCoR': [b, ci, co, r],
               'bnact_apply_us': us(lambda: be.bnact_forward(x3, gamma, beta, None, None, False, 0.0, 0.0, 0.1, stats=(mean, rstd))),
               'absmax_us': us(lambda: be.absmax_bits(act)),
               'bnact_absmax_us': us(lambda: be.bnact_absmax_bits(x, bn)),
               'conv_fwd_plain_us': us(lambda: be.conv3d_igemm_split(act, wts, bias, co, 2, True, am_act)),
               'conv_fwd_folded_us': us(lambda: be.conv3d_igemm_split_bnact(x, wts, bias, co, bn, True, am_x))}
        if be.conv3d_backward_weight_f16_serves(x):
            row['wgrad_plain_us'] = us(lambda: be.conv3d_backward_weight_f16(act, gy, am_act, am_gy, with_bias=True))
            row['wgrad_folded_us'] = us(lambda: be.conv3d_backward_weight_f16_bnact(x, gy, am_x, am_gy, bn, with_bias=True))
        g3 = torch.randn(b, ci, r ** 3, device=dev) * 1e-3
        row['bnact_bwd_us'] = us(lambda: be.bnact_backward(x3, g3, gamma, beta, mean, rstd, 0.1, True))
        row['bnact_bwd_with_amax_us'] = us(lambda: be.bnact_backward(x3, g3, gamma, beta, mean, rstd, 0.1, True, want_amax=True))
        row['absmax_of_grad_us'] = us(lambda: be.absmax_bits(g3))
        row['forward_plain_total_us'] = round(row['bnact_apply_us'] + row['absmax_us'] + row['conv_fwd_plain_us'], 1)
        row['forward_folded_total_us'] = round(row['bnact_absmax_us'] + row['conv_fwd_folded_us'], 1)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
