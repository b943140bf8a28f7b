"""The 256-row persistent 1x1 GEMM (csrc/pointwise_bf16.hip, pw_gemm_f16_wide_kernel: round 6, the default for K % 64 == 0, M >= 256 with an
even number of 128-row blocks, N % 256 == 0) is BIT-IDENTICAL to the 128-row kernel it replaces (PVCNN_PW_WIDE=0, read once per process:
hence two child processes) -- outputs AND the BatchNorm partial sums of the epilogue --, forward and backward-data, incl. a channel
count whose last 256-row block is padded (M = 1472), one item per workgroup (tiny batches), several items per workgroup (persistence:
more items than CUs), per-tile scales that differ by decades from tile to tile (the request streams run one to three tiles ahead of the
multiply: each tile must be converted with ITS item's scale) and K = 64 (one group of steps per item)."""
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
for x, w, bias, gy in cases:
    x, w, bias, gy = x.cuda(), w.cuda(), bias.cuda(), gy.cuda()
    y, part = be.pwconv_forward_split(x, w, bias, 2, want_stats=True)
    y_nb = be.pwconv_forward_split(x, w, None, 2)
    gx = be.pwconv_backward_data_split(gy, w, 2)
    out.append([y.cpu(), part.cpu(), y_nb.cpu(), gx.cpu()])
torch.save(out, sys.argv[3])
'''

# (B, Ci, Co, N): forward is K = Ci, M = Co; backward-data K = Co, M = Ci
SHAPES = [(2, 1472, 512, 2048), (16, 128, 1024, 4096), (3, 512, 256, 1024), (1, 64, 256, 256), (2, 256, 512, 512), (40, 64, 1472, 2048)]


def test_the_wide_persistent_gemm_is_bit_identical_to_the_128_row_kernel(tmp_path):
    g = torch.Generator().manual_seed(17)
    cases = []
    for b, ci, co, n in SHAPES:
        x = torch.randn(b, ci, n, generator=g)
        # every 256-point tile at a scale of its own, decades apart (range contract: per-tile scales)
        x = x * torch.pow(10.0, torch.randint(-6, 7, (b, 1, n // 256), generator=g).float()).repeat_interleave(256, dim=2)
        gy = torch.randn(b, co, n, generator=g) * torch.pow(10.0, torch.randint(-9, 3, (b, 1, n // 256), generator=g).float()).repeat_interleave(256, dim=2)
        cases.append((x, torch.randn(co, ci, generator=g) * 0.1, torch.randn(co, generator=g), gy))
    torch.save(cases, tmp_path / 'cases.pt')
    script = tmp_path / 'child.py'
    script.write_text(_CHILD)
    outs = {}
    # '1': the default (512 x 128 items where the image has a multiple of four 128-row blocks, 256 x 256 items else); '2': 256 x 256 items only
    for tag, flag in (('narrow', '0'), ('wide', '1'), ('wide256', '2')):
        env = dict(os.environ, PVCNN_PW_WIDE=flag)
        subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / 'cases.pt'), str(tmp_path / f'{tag}.pt')], check=True, env=env, timeout=600)
        outs[tag] = torch.load(tmp_path / f'{tag}.pt')
    for tag in ('wide', 'wide256'):
        for case, (a, b_) in enumerate(zip(outs['narrow'], outs[tag])):
            for k, (p, q) in enumerate(zip(a, b_)):
                assert torch.equal(p, q), (tag, SHAPES[case], ['y', 'stats_part', 'y without bias', 'grad_x'][k], (p - q).abs().max().item())
    # and against fp64 on the first case (the kernels agree with each other AND with the truth)
    x, w, bias, _ = cases[0]
    ref = torch.einsum('oc,bcn->bon', w.double(), x.double()) + bias.double().view(1, -1, 1)
    # per tile: the error relative to the tile's own largest output (the range contract's statement)
    err = (outs['wide'][0][0].double() - ref).abs().view(2, 512, -1, 256).amax(dim=(1, 3)) / ref.abs().view(2, 512, -1, 256).amax(dim=(1, 3))
    assert err.max().item() < 1e-5, err.max().item()
