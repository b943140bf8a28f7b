"""The software-pipelined R = 32 devoxelize gather (csrc/slab.h, gather_lds_pipe_kernel: the default; the bench's roofline kernel) is
BIT-IDENTICAL to the classic single-row-slab gather it replaced (PVCNN_GATHER_PIPE=0 -- the one debugging switch the kernels read from the
environment, once per process: hence two child processes), with and without the fused BatchNorm + LeakyReLU + addend."""
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
for case in cases:
    coords, feat, bn, addend = (t.cuda() if t is not None else None for t in case)
    r = 32
    plain = be.trilinear_devoxelize_forward(r, True, coords, feat)
    gamma, beta, mean, rstd = bn.unbind(0)
    fused = be.trilinear_devoxelize_bnact_forward(r, True, coords, feat, gamma, beta, mean, rstd, 0.1, addend)
    out.append([t.cpu() for t in plain] + [t.cpu() for t in fused])
torch.save(out, sys.argv[3])
'''


def test_pipelined_gather_is_bit_identical(tmp_path):
    g = torch.Generator().manual_seed(5)
    cases = []
    for b, c, n in [(16, 64, 4096), (8, 71, 4096), (2, 5, 1024), (1, 1, 4092), (3, 130, 2048)]:
        coords = torch.rand(b, 3, n, generator=g) * 31
        coords[:, :, :n // 8] = torch.round(coords[:, :, :n // 8])            # integral coordinates: zero hi offsets
        coords[:, 2, n // 8:n // 4] = 7.0                                      # a plane
        feat = torch.randn(b, c, 32 ** 3, generator=g)
        bn = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.2,
                          torch.rand(c, generator=g) + 0.5])
        addend = torch.randn(b, c, n, generator=g)
        cases.append((coords, feat, bn, addend))
    torch.save(cases, tmp_path / 'cases.pt')
    script = tmp_path / 'child.py'
    script.write_text(_CHILD)
    outs = {}
    for tag, flag in (('default', '0'), ('pipe', '1')):
        env = dict(os.environ, PVCNN_GATHER_PIPE=flag)
        subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / 'cases.pt'), str(tmp_path / f'{tag}.pt')], check=True, env=env,
                       timeout=300)
        outs[tag] = torch.load(tmp_path / f'{tag}.pt')
    for case, (a, b_) in enumerate(zip(outs['default'], outs['pipe'])):
        for k, (x, y) in enumerate(zip(a, b_)):
            assert torch.equal(x, y), (case, k)
