"""Seeded random-shape sweep over the GEMM-like kernels and the fused gather: shapes nobody hand-picked.

Each case compares against an fp64 torch evaluation (convolutions) or against the unfused native ops (gather)."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rel(a, b):
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def test_pointwise_conv_random_shapes(hip):
    from pvcnn_amd.modules.functional.pwconv import pointwise_conv
    rng = random.Random(20260924)
    for case in range(24):
        b = rng.randint(1, 3)
        ci = rng.choice([1, 3, 9, 31, 32, 33, 64, 67, 100, 131, 200])
        co = rng.choice([1, 4, 13, 32, 48, 64, 96, 128, 160, 260])
        n = rng.choice([1, 5, 64, 100, 256, 512, 1000, 1024, 1280])
        torch.manual_seed(case)
        x = torch.randn(b, ci, n, device=DEV, requires_grad=True)
        w = (torch.randn(co, ci, 1, device=DEV) * 0.2).requires_grad_()
        bias = torch.randn(co, device=DEV, requires_grad=True) if case % 3 else None
        gy = torch.randn(b, co, n, device=DEV)
        y = pointwise_conv(x, w, bias)
        y.backward(gy)
        xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
        bd = bias.detach().double().requires_grad_() if bias is not None else None
        yd = F.conv1d(xd, wd, bd)
        yd.backward(gy.double())
        tag = f'case {case}: B={b} Ci={ci} Co={co} N={n}'
        assert _rel(y.detach(), yd.detach()) < 1e-5, tag
        assert _rel(x.grad, xd.grad) < 1e-5, tag
        assert _rel(w.grad, wd.grad) < 1e-5, tag
        if bias is not None:
            assert _rel(bias.grad, bd.grad) < 1e-5, tag


def test_conv3d_random_shapes(hip):
    from pvcnn_amd.modules.functional.conv3d import voxel_conv3d
    rng = random.Random(1588147245)
    for case in range(16):
        r = rng.choice([2, 3, 4, 6, 8, 9, 12, 16])
        b = rng.randint(1, 3) if r <= 12 else 1
        ci = rng.choice([1, 2, 5, 8, 16, 33])
        co = rng.choice([1, 4, 12, 32, 64, 68, 100])
        torch.manual_seed(100 + case)
        x = torch.randn(b, ci, r, r, r, device=DEV, requires_grad=True)
        w = (torch.randn(co, ci, 3, 3, 3, device=DEV) * 0.1).requires_grad_()
        bias = torch.randn(co, device=DEV, requires_grad=True)
        gy = torch.randn(b, co, r, r, r, device=DEV)
        y = voxel_conv3d(x, w, bias)
        y.backward(gy)
        xd, wd, bd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_(), bias.detach().double().requires_grad_()
        yd = F.conv3d(xd, wd, bd, padding=1)
        yd.backward(gy.double())
        tag = f'case {case}: B={b} Ci={ci} Co={co} R={r}'
        assert _rel(y.detach(), yd.detach()) < 1e-5, tag
        assert _rel(x.grad, xd.grad) < 1e-5, tag
        assert _rel(w.grad, wd.grad) < 1e-5, tag
        assert _rel(bias.grad, bd.grad) < 1e-5, tag


def test_conv_epilogue_statistics_match_a_pass_over_the_output(hip):
    """BatchNorm partial sums emitted by the convolution epilogues vs statistics computed from the stored output."""
    rng = random.Random(7)
    for case in range(8):
        torch.manual_seed(200 + case)
        if case % 2 == 0:
            b, ci, co, r = rng.randint(1, 3), rng.choice([3, 8, 16]), rng.choice([8, 32, 64, 96]), rng.choice([4, 8, 16])
            x = torch.randn(b, ci, r, r, r, device=DEV)
            w = torch.randn(co, ci, 3, 3, 3, device=DEV) * 0.1
            bias = torch.randn(co, device=DEV)
            y, part = hip.conv3d_forward(x, w, bias, want_stats=True)
        else:
            b, ci, co, n = rng.randint(1, 3), rng.choice([9, 32, 67]), rng.choice([13, 64, 160]), rng.choice([100, 256, 1024])
            x = torch.randn(b, ci, n, device=DEV)
            w = torch.randn(co, ci, device=DEV) * 0.1
            bias = torch.randn(co, device=DEV)
            y, part = hip.pwconv_forward(x, w, bias, want_stats=True)
        # the partials are sums of (y - bias) and (y - bias)^2: the shift keeps the variance well conditioned
        yc = (y.double() - bias.double().view(1, co, *([1] * (y.dim() - 2)))).transpose(0, 1).reshape(co, -1)
        sums = part.double().sum(dim=1)
        assert _rel(sums[:, 0], yc.sum(dim=1)) < 1e-5, f'case {case}'
        assert _rel(sums[:, 1], (yc * yc).sum(dim=1)) < 1e-5, f'case {case}'


def test_fused_gather_random_shapes(hip):
    """devoxelize(leaky_relu(bn(grid))) fused vs the separate native ops, bit for bit, over odd shapes."""
    rng = random.Random(99)
    for case in range(12):
        r = rng.choice([2, 3, 4, 5, 8, 12, 16, 32])
        b, c = rng.randint(1, 3), rng.choice([1, 3, 8, 17, 64])
        n = rng.choice([1, 7, 64, 100, 1000, 4096, 5000])
        torch.manual_seed(300 + case)
        grid = torch.randn(b, c, r ** 3, device=DEV)
        coords = torch.rand(b, 3, n, device=DEV) * (r - 1)
        gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
        act, mean, rstd = hip.bnact_forward(grid, gamma, beta, None, None, True, 0.1, 1e-4, 0.1)
        ref = hip.trilinear_devoxelize_forward(r, True, coords, act)
        got = hip.trilinear_devoxelize_bnact_forward(r, True, coords, grid, gamma, beta, mean, rstd, 0.1)
        for a, bb in zip(got, ref):
            assert torch.equal(a, bb), f'case {case}: B={b} C={c} N={n} R={r}'


def _skewed_coords(rng, b, n, r, mode):
    """float grid coordinates in [0, r-1] with the point distributions that stress the per-cloud counting sort:
    uniform, one voxel, two tight clusters, an axis-aligned plane, a line."""
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    c = torch.rand(b, 3, n, generator=g) * (r - 1)
    if mode == 'one_voxel':
        c = torch.full((b, 3, n), float(rng.randint(0, r - 1))) + torch.rand(b, 3, n, generator=g) * 0.25
    elif mode == 'clusters':
        centre = torch.where(torch.rand(b, 1, n, generator=g) < 0.7, 0.2 * (r - 1), 0.8 * (r - 1))
        c = centre + torch.randn(b, 3, n, generator=g) * 0.6
    elif mode == 'plane':
        c[:, rng.randint(0, 2), :] = float(rng.randint(0, r - 1))
    elif mode == 'line':
        c[:, 0, :] = float(rng.randint(0, r - 1))
        c[:, 1, :] = float(rng.randint(0, r - 1))
    return c.clamp(0, r - 1).contiguous()


def test_scatter_ops_match_the_oracle_on_skewed_clouds(hip, oracle):
    """avg_voxelize fwd and trilinear_devoxelize bwd, bit for bit against the serial point-order oracle, on the
    distributions that unbalance the entry-balanced CSR ranges (and their list-overflow path)."""
    rng = random.Random(4242)
    for case, mode in enumerate(['uniform', 'one_voxel', 'clusters', 'plane', 'line', 'plane', 'clusters', 'one_voxel']):
        r = rng.choice([4, 8, 12, 16, 32])
        b, c = rng.randint(1, 3), rng.choice([1, 5, 16, 64])
        n = rng.choice([33, 1000, 4096, 6000])
        torch.manual_seed(400 + case)
        norm = _skewed_coords(rng, b, n, r, mode)
        vox = torch.round(norm).to(torch.int32)
        feat = torch.randn(b, c, n)
        tag = f'case {case} ({mode}): B={b} C={c} N={n} R={r}'
        o = oracle.avg_voxelize_forward(feat, vox, r)
        h = hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), r)
        for a, e in zip(h, o):
            assert torch.equal(a.cpu(), e), tag
        grid = torch.randn(b, c, r ** 3)
        _, o_inds, o_wgts = oracle.trilinear_devoxelize_forward(r, True, norm, grid)
        _, h_inds, h_wgts = hip.trilinear_devoxelize_forward(r, True, norm.to(DEV), grid.to(DEV))
        assert torch.equal(h_inds.cpu(), o_inds) and torch.equal(h_wgts.cpu(), o_wgts), tag
        gy = torch.randn(b, c, n)
        assert torch.equal(hip.trilinear_devoxelize_backward(gy.to(DEV), h_inds, h_wgts, r).cpu(),
                           oracle.trilinear_devoxelize_backward(gy, o_inds, o_wgts, r)), tag
