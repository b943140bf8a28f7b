"""Kernel variants selected by environment variable, compared across PROCESSES (the switch is read once per process).  Opt-in:
`PVCNN_EXPERIMENTAL=1 python -m pytest tests/test_gpu_experimental.py -m gpu` (the in-process equivalents that ran on the GPU
are tools/pipecheck.py / pipecheck2.py: profiles/ab/r02_pipecheck_*).

  PVCNN_GATHER_PIPE=0 | 1   classic | software-pipelined (default) single-row-slab gather (csrc/slab.h, gather_lds_pipe_kernel):
                            trilinear_devoxelize forward at R = 32 (the bench's roofline kernel), with and without the fused
                            BatchNorm + LeakyReLU + addend.  Must be BIT-IDENTICAL.
  PVCNN_AMAX_REDUCE=2       second form of the gradient maximum that rides on the BatchNorm backward's apply pass (csrc/bnact.hip,
                            block_atomic_max_bits_v2: LDS-only barrier, filter value read at kernel start, fire-and-forget
                            atomic).  Must give the same bits as pvcnn_absmax_bits of the gradient; time it with tools/foldbench.py.
  PVCNN_PW_MB8=1            256-channel workgroup tile of the f16x2 1x1 GEMM for M % 256 == 0 (csrc/pointwise_bf16.hip, MB = 8: one
                            workgroup per CU, accumulators in AGPRs).  Same products in the same order: BIT-IDENTICAL to the 128-channel
                            tile, statistics partials included; time it with `PVCNN_PW_MB8=1 python tools/pwbench.py`.
  PVCNN_WGRAD_REDUCE=2      four threads per element in the split-K reduction of the Conv3d backward-weight (csrc/conv3d_wgrad_f16.hip,
                            conv3d_wgrad_f16_reduce_v2_kernel).  Deterministic, within 1e-5 of fp64, NOT bit-identical to the first
                            form (another summation tree); time: `PVCNN_WGRAD_REDUCE=2 python tools/convcheck.py --time --no-check`.
"""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('PVCNN_EXPERIMENTAL') != '1', reason='experimental variants: set PVCNN_EXPERIMENTAL=1')]

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


_CHILD_AMAX = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
torch.manual_seed(3)
ok = True
for b, c, s in [(16, 64, 32 ** 3), (2, 64, 4096), (3, 24, 1000), (2, 7, 1001), (16, 128, 16 ** 3)]:
    x = torch.randn(b, c, s, device='cuda')
    g = torch.randn(b, c, s, device='cuda') * 1e-3
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    mean, rstd = torch.randn(c, device='cuda') * 0.3, torch.rand(c, device='cuda') + 0.5
    for training in (True, False):
        for scale in (1.0, 0.5):
            gx, gg, gb, amax = be.bnact_backward(x, g * scale, gamma, beta, mean, rstd, 0.1, training, want_amax=True)
            plain = be.bnact_backward(x, g * scale, gamma, beta, mean, rstd, 0.1, training)
            ok = ok and torch.equal(gx, plain[0]) and torch.equal(amax, be.absmax_bits(gx))
print('AMAX_OK' if ok else 'AMAX_MISMATCH')
"""


def test_second_form_of_the_gradient_maximum(tmp_path):
    script = tmp_path / 'child_amax.py'
    script.write_text(_CHILD_AMAX)
    for form in ('1', '2'):
        out = subprocess.run([sys.executable, str(script), ROOT], check=True, env=dict(os.environ, PVCNN_AMAX_REDUCE=form), timeout=300,
                             capture_output=True, text=True).stdout
        assert 'AMAX_OK' in out, (form, out)


_CHILD_PW = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
torch.manual_seed(4)
out = []
for b, k, m, n in [(16, 1472, 512, 4096), (2, 512, 256, 4096), (3, 128, 1024, 1000), (1, 40, 256, 260), (2, 64, 768, 512)]:
    x = torch.randn(b, k, n, device='cuda')
    w = torch.randn(m, k, device='cuda') * 0.05
    bias = torch.randn(m, device='cuda')
    y, part = be.pwconv_forward_split(x, w, bias, 2, want_stats=True)
    gx = be.pwconv_backward_data_split(torch.randn(b, m, n, device='cuda', generator=torch.Generator('cuda').manual_seed(9)) * 1e-3, w, 2)
    ref = torch.einsum('mk,bkn->bmn', w.double(), x.double()) + bias.double().view(1, -1, 1)
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    out.append((y.cpu(), part.cpu(), gx.cpu(), err))
torch.save(out, sys.argv[2])
"""


def test_256_channel_tile_of_the_pointwise_gemm(tmp_path):
    script = tmp_path / 'child_pw.py'
    script.write_text(_CHILD_PW)
    res = {}
    for flag in ('0', '1'):
        subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / f'pw{flag}.pt')], check=True, env=dict(os.environ, PVCNN_PW_MB8=flag),
                       timeout=300)
        res[flag] = torch.load(tmp_path / f'pw{flag}.pt')
    for case, (a, b_) in enumerate(zip(res['0'], res['1'])):
        assert b_[3] < 1e-5, (case, b_[3])
        for k in range(3):
            assert torch.equal(a[k], b_[k]), (case, k)


_CHILD_WG = r"""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, sys.argv[1])
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
g = torch.Generator().manual_seed(7)
out = []
for b, ci, co, r in [(2, 9, 64, 32), (16, 64, 64, 16), (3, 40, 70, 16), (1, 33, 130, 32), (1, 1, 1, 16), (20, 64, 128, 16)]:
    x = torch.randn(b, ci, r, r, r, generator=g).cuda()
    gy = (torch.randn(b, co, r, r, r, generator=g) * 1e-3).cuda()
    wd = torch.zeros(co, ci, 3, 3, 3, device='cuda', dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), wd, padding=1).backward(gy.double())
    gw, gb = be.conv3d_backward_weight_f16(x, gy, with_bias=True)
    gw2, gb2 = be.conv3d_backward_weight_f16(x, gy, with_bias=True)
    err = ((gw.double() - wd.grad).abs().max() / wd.grad.abs().max()).item()
    out.append((gw.cpu(), gb.cpu(), err, bool(torch.equal(gw, gw2) and torch.equal(gb, gb2))))
torch.save(out, sys.argv[2])
"""


def test_second_form_of_the_backward_weight_reduction(tmp_path):
    script = tmp_path / 'child_wg.py'
    script.write_text(_CHILD_WG)
    res = {}
    for form in ('1', '2'):
        subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / f'wg{form}.pt')], check=True,
                       env=dict(os.environ, PVCNN_WGRAD_REDUCE=form), timeout=300)
        res[form] = torch.load(tmp_path / f'wg{form}.pt')
    for case, (a, b_) in enumerate(zip(res['1'], res['2'])):
        assert b_[2] < 1e-5 and b_[3], (case, b_[2], b_[3])              # within the fp64 bar, run-to-run identical
        assert torch.equal(a[1], b_[1]), case                             # grad_bias: same order in both forms
        scale = a[0].abs().max().item()
        assert (a[0] - b_[0]).abs().max().item() <= 2e-6 * scale, case     # another summation tree, same sum
