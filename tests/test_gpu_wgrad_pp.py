"""The round-6 Conv3d backward-weight kernel (csrc/conv3d_wgrad_f16.hip, conv3d_wgrad_f16_pp_kernel, the default) next to the kernel it
replaces (PVCNN_WGRAD_PP=0, read once per process: child processes).  It applies the z shift of a tap to the grad_y fragment instead of
the x fragment -- the same products, grouped into other 16-deep k-steps --, so grad_w agrees to fp32 rounding of the sums (<= 2e-6 of
the largest weight gradient; both are held to 1e-5 against fp64), grad_bias (the same additions in the same order) bit for bit, and
two runs of the new kernel bit for bit (split-K partials summed in a fixed order: deterministic).
Cases: every grid the kernel serves (R = 8, 12, 16, 32), the packed first layer (Ci = 9), channel counts with padded blocks, a batch
with fewer strips than partitions and one with a ragged last pass, a voxelised cloud with an amax buffer (zero rows skipped), rows
decades apart."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

_CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
cases = torch.load(sys.argv[2])
out = []
for x, gy, with_table in cases:
    x, gy = x.cuda(), gy.cuda()
    gw, gb = be.conv3d_backward_weight_f16(x, gy, x_amax=be.conv_amax(x) if with_table else None, with_bias=True)
    out.append([gw.cpu(), gb.cpu()])
torch.save(out, sys.argv[3])
'''

SHAPES = [(16, 64, 64, 32), (2, 9, 64, 32), (16, 64, 64, 16), (16, 128, 128, 16), (5, 64, 128, 16), (3, 9, 64, 16), (1, 40, 72, 16),   # (B, Ci, Co, R)
          (7, 33, 130, 12), (32, 64, 64, 12), (3, 16, 48, 8), (1, 8, 8, 8)]


def test_the_round6_wgrad_kernel_next_to_the_kernel_it_replaces(tmp_path):
    g = torch.Generator().manual_seed(29)
    cases = []
    for k, (b, ci, co, r) in enumerate(SHAPES):
        x = torch.randn(b, ci, r, r, r, generator=g) * torch.pow(10.0, torch.randint(-3, 3, (b, 1, r, r, 1), generator=g).float())
        with_table = k in (1, 5, 6)
        if with_table:                                            # a voxelised cloud: most z rows empty, the amax buffer says which
            keep = torch.zeros(b, 1, r, r, 1)
            keep[:, :, 9 * r // 32:23 * r // 32, 9 * r // 32:23 * r // 32] = 1.0
            x = x * keep
        gy = torch.randn(b, co, r, r, r, generator=g) * torch.pow(10.0, torch.randint(-4, 2, (b, 1, r, r, 1), generator=g).float())
        cases.append((x, gy, with_table))
    torch.save(cases, tmp_path / 'cases.pt')
    script = tmp_path / 'child.py'
    script.write_text(_CHILD)
    outs = {}
    for tag, flag in (('old', '0'), ('pp', '1'), ('pp_again', '1')):
        subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / 'cases.pt'), str(tmp_path / f'{tag}.pt')], check=True,
                       env=dict(os.environ, PVCNN_WGRAD_PP=flag), timeout=900)
        outs[tag] = torch.load(tmp_path / f'{tag}.pt')
    for case, (a, b_, c) in enumerate(zip(outs['old'], outs['pp'], outs['pp_again'])):
        assert torch.equal(b_[0], c[0]) and torch.equal(b_[1], c[1]), (SHAPES[case], 'two runs differ')
        assert torch.equal(a[1], b_[1]), (SHAPES[case], 'grad_bias')
        assert ((a[0] - b_[0]).abs().max() / a[0].abs().max()).item() < 2e-6, (SHAPES[case], 'grad_w')
    # ... and with the truth (fp64), relative to the largest weight gradient: the kernel's contract (include/pvcnn_hip.h)
    for case, (x, gy, _) in enumerate(cases):
        w = torch.zeros(SHAPES[case][2], SHAPES[case][1], 3, 3, 3, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv3d(x.double(), w, padding=1).backward(gy.double())
        gw, gb = outs['pp'][case]
        assert ((gw.double() - w.grad).abs().max() / w.grad.abs().max()).item() < 1e-5, SHAPES[case]
        gbr = gy.double().sum(dim=(0, 2, 3, 4))
        assert ((gb.double() - gbr).abs().max() / gy.double().abs().sum(dim=(0, 2, 3, 4)).max()).item() < 1e-5, SHAPES[case]
