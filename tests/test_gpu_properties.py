"""BASELINE.json's FULL sizes -- (B,C,N,R) = (16,64,4096,32) and (16,128,4096,16), the PVConv stages of configs[1]:
bit-exact parity with the oracle (it finishes these in seconds: 33 M multiply-adds per op) and the
size-independent properties of the voxelize / devoxelize pair -- adjointness, linearity, partition of unity, mass
conservation, run-to-run determinism.  Reference semantics: vox.cu:18-110, trilinear_devox.cu:21-162."""
import pytest
import torch

from conftest import synth_cloud

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
FULL = [(16, 64, 4096, 32), (16, 128, 4096, 16)]
# BASELINE configs[3] (ShapeNet PVCNN, B=64 N=2048: 64 ch @ R=32, 128 ch @ R=16) and configs[4] (Frustum-PVCNN, B=32
# N=1024, R=12 -- a non-power-of-two grid -- and R=16), SURVEY.md 8(d) op-level shapes
FULL_OTHER = [(64, 128, 2048, 16), (64, 64, 2048, 32), (32, 64, 1024, 12), (32, 128, 1024, 12), (32, 64, 1024, 16)]


def _inputs(b, c, n, r, kind, seed):
    g = torch.Generator().manual_seed(seed)
    co = synth_cloud(g, b, n, kind)
    co = co / co.amax(dim=(1, 2), keepdim=True).clamp(min=1e-6)
    norm = torch.clamp(co * r, 0, r - 1).contiguous()
    vox = torch.round(norm).to(torch.int32).contiguous()
    return g, norm, vox


@pytest.mark.parametrize('b,c,n,r,kind', [(*f, k) for f in FULL for k in ('cube', 'surface')] +
                         [(*f, 'cube') for f in FULL_OTHER] + [(32, 64, 1024, 12, 'surface')])
def test_full_size_ops_equal_the_oracle(hip, oracle, b, c, n, r, kind):
    g, norm, vox = _inputs(b, c, n, r, kind, 1588147245)
    feat = torch.randn(b, c, n, generator=g)
    grid = torch.randn(b, c, r ** 3, generator=g)
    o_out, o_ind, o_cnt = oracle.avg_voxelize_forward(feat, vox, r)
    h_out, h_ind, h_cnt = hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), r)
    assert torch.equal(h_ind.cpu(), o_ind) and torch.equal(h_cnt.cpu(), o_cnt) and torch.equal(h_out.cpu(), o_out)
    assert torch.equal(hip.avg_voxelize_backward(grid.to(DEV), h_ind, h_cnt).cpu(), oracle.avg_voxelize_backward(grid, o_ind, o_cnt))
    o_pts, o_inds, o_wgts = oracle.trilinear_devoxelize_forward(r, True, norm, grid)
    h_pts, h_inds, h_wgts = hip.trilinear_devoxelize_forward(r, True, norm.to(DEV), grid.to(DEV))
    assert torch.equal(h_inds.cpu(), o_inds) and torch.equal(h_wgts.cpu(), o_wgts) and torch.equal(h_pts.cpu(), o_pts)
    assert torch.equal(hip.trilinear_devoxelize_backward(feat.to(DEV), h_inds, h_wgts, r).cpu(),
                       oracle.trilinear_devoxelize_backward(feat, o_inds, o_wgts, r))


@pytest.mark.parametrize('b,c,n,r', FULL)
def test_full_size_properties(hip, b, c, n, r):
    g, norm, vox = _inputs(b, c, n, r, 'cube', 7)
    norm, vox = norm.to(DEV), vox.to(DEV)
    f1, f2 = torch.randn(b, c, n, generator=g).to(DEV), torch.randn(b, c, n, generator=g).to(DEV)
    g1, g2 = torch.randn(b, c, r ** 3, generator=g).to(DEV), torch.randn(b, c, r ** 3, generator=g).to(DEV)

    # devoxelize: adjoint pair <D g, f> = <g, D^T f>, linearity, partition of unity (constant grid -> constant)
    pts, inds, wgts = hip.trilinear_devoxelize_forward(r, True, norm, g1)
    back = hip.trilinear_devoxelize_backward(f1, inds, wgts, r)
    lhs, rhs = (pts.double() * f1.double()).sum(), (g1.double() * back.double()).sum()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0) * 10
    lin = hip.trilinear_devoxelize_forward(r, True, norm, 0.5 * g1 + g2)[0]
    sep = 0.5 * pts + hip.trilinear_devoxelize_forward(r, True, norm, g2)[0]
    assert torch.allclose(lin, sep, rtol=1e-5, atol=1e-5)
    ones = hip.trilinear_devoxelize_forward(r, True, norm, torch.full_like(g1, 3.0))[0]
    assert torch.allclose(ones, torch.full_like(ones, 3.0), rtol=0, atol=3e-6)
    assert torch.allclose(wgts.sum(dim=1), torch.ones(b, n, device=DEV), atol=1e-6)

    # voxelize: adjoint pair, mass conservation (sum_v out[v] * cnt[v] = sum_i f[i]), counts add up to N
    out, ind, cnt = hip.avg_voxelize_forward(f1, vox, r)
    backv = hip.avg_voxelize_backward(g1, ind, cnt)
    lhs, rhs = (out.double() * g1.double()).sum(), (f1.double() * backv.double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    assert torch.equal(cnt.sum(dim=1), torch.full((b,), n, dtype=cnt.dtype, device=DEV))
    mass = (out.double() * cnt.unsqueeze(1).double()).sum(dim=2)
    assert torch.allclose(mass, f1.double().sum(dim=2), rtol=1e-5, atol=1e-3)
    lin = hip.avg_voxelize_forward(0.5 * f1 + f2, vox, r)[0]
    sep = 0.5 * out + hip.avg_voxelize_forward(f2, vox, r)[0]
    assert torch.allclose(lin, sep, rtol=1e-5, atol=1e-5)

    # run-to-run determinism of the scatters at full size
    assert torch.equal(hip.avg_voxelize_forward(f1, vox, r)[0], out)
    assert torch.equal(hip.trilinear_devoxelize_backward(f1, inds, wgts, r), back)
