"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors, so these analytic identities -- each derived by
hand from the cited reference lines -- are the first pin of oracle/pvcnn_oracle.c.  (The second
pin is oracle/_ref: the reference's own kernels run on the CPU, tests/test_oracle_vs_ref.py.)
"""
import math

import pytest
import torch

from conftest import grid_coords, synth_cloud


def test_devox_constant_grid(oracle, gen):
    # weights sum to 1 (trilinear_devox.cu:45-59) -> a constant grid interpolates to the constant
    r, b, c, n = 8, 2, 3, 200
    feat = torch.full((b, c, r ** 3), 2.5)
    outs, _, wg = oracle.trilinear_devoxelize_forward(r, True, grid_coords(gen, b, n, r), feat)
    assert torch.allclose(wg.sum(1), torch.ones(b, n), atol=1e-6)
    assert torch.allclose(outs, torch.full_like(outs, 2.5), atol=1e-5)


def test_devox_linear_field(oracle, gen):
    # trilinear interpolation reproduces a*x + b*y + c*z + d exactly (up to fp32 rounding)
    r, b, n = 12, 1, 300   # r = 12: not a power of two (Frustum-PVCNN uses it)
    xs = torch.arange(r, dtype=torch.float32)
    field = (0.5 * xs.view(r, 1, 1) - 1.25 * xs.view(1, r, 1) + 2.0 * xs.view(1, 1, r) + 3.0).reshape(1, 1, -1)
    co = grid_coords(gen, b, n, r)
    outs, _, _ = oracle.trilinear_devoxelize_forward(r, False, co, field.contiguous())
    want = 0.5 * co[:, 0] - 1.25 * co[:, 1] + 2.0 * co[:, 2] + 3.0
    assert torch.allclose(outs[:, 0], want, atol=1e-4)


def test_devox_integer_and_half_coords(oracle):
    r = 4
    feat = torch.arange(r ** 3, dtype=torch.float32).view(1, 1, -1)
    co = torch.tensor([[[1.0, 0.5], [2.0, 0.5], [3.0, 0.5]]])   # point 0 integral, point 1 = (.5,.5,.5)
    outs, inds, wgts = oracle.trilinear_devoxelize_forward(r, True, co, feat)
    # integral coordinates: all 8 indices equal idx000, weight 1 on corner 000 (:64-75)
    assert inds[0, :, 0].tolist() == [1 * 16 + 2 * 4 + 3] * 8
    assert wgts[0, :, 0].tolist() == [1.0, 0, 0, 0, 0, 0, 0, 0]
    assert outs[0, 0, 0].item() == 27.0
    # (0.5,0.5,0.5): all weights 1/8, indices {0,1,R,R+1,R^2,R^2+1,R^2+R,R^2+R+1}
    assert wgts[0, :, 1].tolist() == [0.125] * 8
    assert inds[0, :, 1].tolist() == [0, 1, r, r + 1, r * r, r * r + 1, r * r + r, r * r + r + 1]
    assert outs[0, 0, 1].item() == pytest.approx(sum(inds[0, :, 1].tolist()) / 8)


def test_devox_eval_returns_dummies(oracle, gen):
    outs, inds, wgts = oracle.trilinear_devoxelize_forward(4, False, grid_coords(gen, 1, 10, 4), torch.rand(1, 2, 64))
    assert inds.shape == (1,) and wgts.shape == (1,) and outs.shape == (1, 2, 10)   # trilinear_devox.cpp:45-53


def test_voxelize_two_points_one_voxel(oracle):
    r = 2
    feat = torch.tensor([[[1.0, 3.0, 10.0]]])                       # (1,1,3)
    coords = torch.tensor([[[1, 1, 0], [0, 0, 1], [1, 1, 1]]], dtype=torch.int32)   # voxels 5, 5, 3
    out, ind, cnt = oracle.avg_voxelize_forward(feat, coords, r)
    assert ind.tolist() == [[5, 5, 3]]                              # x*r^2 + y*r + z (vox.cu:31)
    assert cnt[0].tolist() == [0, 0, 0, 1, 0, 2, 0, 0]
    assert out[0, 0].tolist() == [0, 0, 0, 10.0, 0, 2.0, 0, 0]      # mean; empty voxels stay 0


def test_voxelize_premultiplied_addends(oracle):
    # vox.cu:66-68 multiplies every addend by 1/cnt BEFORE summing: (a/3 + b/3) + c/3, not (a+b+c)/3
    a, b_, c = 0.1, 0.2, 0.7
    feat = torch.tensor([[[a, b_, c]]], dtype=torch.float32)
    coords = torch.zeros(1, 3, 3, dtype=torch.int32)
    out, _, _ = oracle.avg_voxelize_forward(feat, coords, 1)
    third = torch.tensor(1.0 / 3.0, dtype=torch.float32)
    f = feat[0, 0]
    want = (f[0] * third + f[1] * third) + f[2] * third
    assert out[0, 0, 0].item() == want.item()


def test_adjoint_identities(oracle, gen):
    # <vox(f), G> == <f, vox_bwd(G)> and <devox(V), g> == <V, devox_bwd(g)>
    r, b, c, n = 6, 2, 4, 150
    f = torch.randn(b, c, n, generator=gen)
    vc = torch.randint(0, r, (b, 3, n), generator=gen, dtype=torch.int32)
    out, ind, cnt = oracle.avg_voxelize_forward(f, vc, r)
    G = torch.randn(b, c, r ** 3, generator=gen)
    gx = oracle.avg_voxelize_backward(G, ind, cnt)
    assert (out.double() * G.double()).sum().item() == pytest.approx((f.double() * gx.double()).sum().item(), rel=1e-5)
    V = torch.randn(b, c, r ** 3, generator=gen)
    co = grid_coords(gen, b, n, r)
    outs, inds, wgts = oracle.trilinear_devoxelize_forward(r, True, co, V)
    g = torch.randn(b, c, n, generator=gen)
    gV = oracle.trilinear_devoxelize_backward(g, inds, wgts, r)
    assert (outs.double() * g.double()).sum().item() == pytest.approx((V.double() * gV.double()).sum().item(), rel=1e-5)


def test_ball_query_semantics(oracle):
    # strict '<', first hit pads, no hit -> zeros (ball_query.cu:39-47, ball_query.cpp:20-22)
    pts = torch.tensor([[[0.0, 1.0, 0.5, 0.25, 5.0], [0.0] * 5, [0.0] * 5]])   # x = 0, 1, .5, .25, 5
    ctr = torch.tensor([[[0.0, 5.0, 100.0], [0.0] * 3, [0.0] * 3]])
    idx = oracle.ball_query(ctr, pts, 0.5, 4)
    # centre 0: d = 0, 1, .5 (NOT < .5), .25 -> hits 0, 3; padded with first hit 0
    assert idx[0, 0].tolist() == [0, 3, 0, 0]
    assert idx[0, 1].tolist() == [4, 4, 4, 4]      # only point 4
    assert idx[0, 2].tolist() == [0, 0, 0, 0]      # no hit: row stays zero
    # more hits than slots: first U in index order, scan stops
    idx = oracle.ball_query(ctr, pts, 2.0, 2)
    assert idx[0, 0].tolist() == [0, 1]


def test_fps_collinear(oracle):
    # points on a line at x = 0..8: start at 0, then the far end, then the middle ...
    x = torch.arange(9, dtype=torch.float32)
    co = torch.stack([x, torch.zeros(9), torch.zeros(9)]).unsqueeze(0).contiguous()
    idx = oracle.furthest_point_sampling(co, 5)
    assert idx[0, :3].tolist() == [0, 8, 4]
    assert sorted(idx[0, 3:].tolist()) == [2, 6]
    # tie (2 and 6 both at distance 2): slot rule = lowest k since both < 512
    assert idx[0, 3].item() == 2


def test_fps_tie_rule_slots(oracle):
    # equidistant candidates k=5 and k=512+3: slot 3 < slot 5, so the HIGHER index 515 wins
    n = 600
    co = torch.zeros(1, 3, n)
    co[0, 0, 5] = 1.0
    co[0, 0, 515] = 1.0
    idx = oracle.furthest_point_sampling(co.contiguous(), 2)
    assert idx[0].tolist() == [0, 515]


def test_three_nn_equidistant(oracle):
    # query at the centroid of an equilateral triangle: weights 1/3 each, indices in scan order
    h = math.sqrt(3) / 2
    ctr = torch.tensor([[[0.0, 1.0, 0.5], [0.0, 0.0, h], [0.0, 0.0, 0.0]]])
    pts = torch.tensor([[[0.5], [h / 3], [0.0]]])
    feat = torch.tensor([[[3.0, 6.0, 9.0]]])
    out, idx, w = oracle.three_nearest_neighbors_interpolate_forward(pts, ctr, feat)
    assert sorted(idx[0, :, 0].tolist()) == [0, 1, 2]
    assert torch.allclose(w[0, :, 0], torch.full((3,), 1 / 3), atol=1e-6)
    assert out[0, 0, 0].item() == pytest.approx(6.0, abs=1e-5)


def test_three_nn_fewer_than_three_centres(oracle):
    # M = 1: unused slots keep index 0 and clamp to d^2 = 1e10 (neighbor_interpolate.cu:37-63)
    ctr = torch.tensor([[[0.0], [0.0], [0.0]]])
    pts = torch.tensor([[[1.0], [0.0], [0.0]]])
    out, idx, w = oracle.three_nearest_neighbors_interpolate_forward(pts, ctr, torch.tensor([[[4.0]]]))
    assert idx[0, :, 0].tolist() == [0, 0, 0]
    assert w[0, :, 0].sum().item() == pytest.approx(1.0, abs=1e-6)
    assert out[0, 0, 0].item() == pytest.approx(4.0, abs=1e-5)


def test_grouping_gather_roundtrip(oracle, gen):
    b, c, n, m, u = 2, 3, 50, 7, 4
    f = torch.randn(b, c, n, generator=gen)
    idx = torch.randint(0, n, (b, m, u), generator=gen, dtype=torch.int32)
    out = oracle.grouping_forward(f, idx)
    want = torch.gather(f.unsqueeze(2).expand(b, c, m, n), 3, idx.long().unsqueeze(1).expand(b, c, m, u))
    assert torch.equal(out, want)
    g = torch.randn(b, c, m, u, generator=gen)
    gx = oracle.grouping_backward(g, idx, n)
    ref = torch.zeros(b, c, n).scatter_add_(2, idx.long().view(b, 1, -1).expand(b, c, m * u), g.view(b, c, -1))
    assert torch.allclose(gx, ref, atol=1e-5)
    gi = torch.randint(0, n, (b, m), generator=gen, dtype=torch.int32)
    assert torch.equal(oracle.gather_features_forward(f, gi), torch.gather(f, 2, gi.long().unsqueeze(1).expand(b, c, m)))


def test_input_contract(oracle):
    # CHECK_CONTIGUOUS / CHECK_IS_FLOAT / CHECK_IS_INT -> RuntimeError (utils.hpp:7-18)
    f = torch.rand(1, 2, 8)
    with pytest.raises(RuntimeError):
        oracle.avg_voxelize_forward(f, torch.zeros(1, 3, 8, dtype=torch.int64), 2)
    with pytest.raises(RuntimeError):
        oracle.avg_voxelize_forward(f.double(), torch.zeros(1, 3, 8, dtype=torch.int32), 2)
    with pytest.raises(RuntimeError):
        oracle.grouping_forward(torch.rand(1, 8, 2).transpose(1, 2), torch.zeros(1, 2, 2, dtype=torch.int32))
