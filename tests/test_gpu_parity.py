"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (include/pvcnn_hip.h "Numerics contract"):
  torch.equal  : EVERYTHING on the default paths -- all int32 outputs, all gathers, and all
                 scatter-adds (avg_voxelize fwd, devoxelize bwd, grouping/gather bwd, 3-NN bwd): the
                 HIP path sums in the oracle's serial point-index order without float atomics;
  atol 1e-5 (+ rtol 1e-5): only the atomic fallback (more than 2^20 scatter targets per cloud: R > 101),
                 whose order is undefined exactly like the reference's atomicAdd.
Sizes: oracle-in-seconds cases here; BASELINE.json's full sizes are covered by the
size-independent properties in test_gpu_properties.py.
"""
import pytest
import torch

from conftest import grid_coords, synth_cloud

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# (B, C, N, R): odd / ragged sizes, non-power-of-two R (Frustum uses 12), N not a multiple of 4
# (scalar kernels), tiny clouds (PVCNN++ deepest level), BASELINE shapes at reduced batch.
VOX_CASES = [(2, 9, 4096, 32), (2, 64, 4096, 16), (3, 5, 1000, 12), (1, 3, 37, 5), (2, 16, 64, 8),
             (1, 130, 2048, 16), (2, 4, 8192, 32), (1, 2, 1, 2)]


def _vox_inputs(gen, b, c, n, r, kind='cube'):
    feat = torch.randn(b, c, n, generator=gen)
    co = synth_cloud(gen, b, n, kind)
    co = co / co.amax(dim=(1, 2), keepdim=True).clamp(min=1e-6)
    norm = torch.clamp(co * r, 0, r - 1)
    return feat, torch.round(norm).to(torch.int32).contiguous(), norm.contiguous()


@pytest.mark.parametrize('b,c,n,r', VOX_CASES)
@pytest.mark.parametrize('kind', ['cube', 'surface'])
def test_avg_voxelize_fwd_bwd(hip, oracle, gen, b, c, n, r, kind):
    feat, vox, _ = _vox_inputs(gen, b, c, n, r, kind)
    o_out, o_ind, o_cnt = oracle.avg_voxelize_forward(feat, vox, r)
    h_out, h_ind, h_cnt = hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), r)
    assert torch.equal(h_ind.cpu(), o_ind)
    assert torch.equal(h_cnt.cpu(), o_cnt)
    assert torch.equal(h_out.cpu(), o_out), 'deterministic CSR voxelize must equal the point-order oracle bit for bit'
    gy = torch.randn(b, c, r ** 3, generator=gen)
    assert torch.equal(hip.avg_voxelize_backward(gy.to(DEV), h_ind, h_cnt).cpu(), oracle.avg_voxelize_backward(gy, o_ind, o_cnt))


def test_avg_voxelize_all_points_in_one_voxel(hip, oracle, gen):
    # degenerate cloud: one voxel holds every point (the stable-rank pass is O(N^2/threads), never serial)
    b, c, n, r = 2, 8, 4096, 16
    feat = torch.randn(b, c, n, generator=gen)
    vox = torch.full((b, 3, n), 7, dtype=torch.int32)
    o = oracle.avg_voxelize_forward(feat, vox, r)
    h = hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), r)
    for a, e in zip(h, o):
        assert torch.equal(a.cpu(), e)


def test_scatters_are_run_to_run_deterministic(hip, gen):
    feat, vox, norm = _vox_inputs(gen, 4, 32, 4096, 16, 'surface')
    a = hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), 16)[0]
    _, inds, wgts = hip.trilinear_devoxelize_forward(16, True, norm.to(DEV), a)
    gb = hip.trilinear_devoxelize_backward(feat.to(DEV), inds, wgts, 16)
    for _ in range(3):
        assert torch.equal(hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), 16)[0], a)
        assert torch.equal(hip.trilinear_devoxelize_backward(feat.to(DEV), inds, wgts, 16), gb)


@pytest.mark.parametrize('r', [40, 104])
def test_avg_voxelize_large_resolution(hip, oracle, gen, r):
    # R = 40: the voxel range of a cloud is split over several workgroups (still the deterministic CSR path, bit-exact);
    # R = 104: more than 2^20 voxels -> atomic fallback (tolerance, like the reference)
    b, c, n = 1, 3, 2000
    feat, vox, _ = _vox_inputs(gen, b, c, n, r)
    o_out, o_ind, o_cnt = oracle.avg_voxelize_forward(feat, vox, r)
    h_out, h_ind, h_cnt = hip.avg_voxelize_forward(feat.to(DEV), vox.to(DEV), r)
    assert torch.equal(h_ind.cpu(), o_ind) and torch.equal(h_cnt.cpu(), o_cnt)
    if r ** 3 <= (1 << 20):
        assert torch.equal(h_out.cpu(), o_out)
    else:
        assert torch.allclose(h_out.cpu(), o_out, atol=1e-5, rtol=1e-5)
    gy = torch.randn(b, c, r ** 3, generator=gen)
    assert torch.equal(hip.avg_voxelize_backward(gy.to(DEV), h_ind, h_cnt).cpu(), oracle.avg_voxelize_backward(gy, o_ind, o_cnt))


@pytest.mark.parametrize('b,c,n,r', VOX_CASES + [(1, 2, 500, 40)])
@pytest.mark.parametrize('training', [True, False])
def test_trilinear_devox_fwd(hip, oracle, gen, b, c, n, r, training):
    feat = torch.randn(b, c, r ** 3, generator=gen)
    co = grid_coords(gen, b, n, r)
    o_outs, o_inds, o_wgts = oracle.trilinear_devoxelize_forward(r, training, co, feat)
    h_outs, h_inds, h_wgts = hip.trilinear_devoxelize_forward(r, training, co.to(DEV), feat.to(DEV))
    assert torch.equal(h_outs.cpu(), o_outs)
    assert h_inds.shape == o_inds.shape and h_wgts.shape == o_wgts.shape
    if training:
        assert torch.equal(h_inds.cpu(), o_inds)
        assert torch.equal(h_wgts.cpu(), o_wgts)


@pytest.mark.parametrize('b,c,n,r', VOX_CASES + [(1, 2, 500, 40)])
def test_trilinear_devox_bwd(hip, oracle, gen, b, c, n, r):
    co = grid_coords(gen, b, n, r)
    _, inds, wgts = oracle.trilinear_devoxelize_forward(r, True, co, torch.zeros(b, 1, r ** 3))
    gy = torch.randn(b, c, n, generator=gen)
    want = oracle.trilinear_devoxelize_backward(gy, inds, wgts, r)
    truth = oracle.trilinear_devoxelize_backward_f64(gy, inds, wgts, r)
    got = hip.trilinear_devoxelize_backward(gy.to(DEV), inds.to(DEV), wgts.to(DEV), r).cpu()
    if r ** 3 <= (1 << 20):   # kCsrMaxTargets
        assert torch.equal(got, want), 'CSR scatter must reproduce the serial (point, corner) order bit for bit'
    else:   # atomic fallback: undefined order, as in the reference
        assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)
        assert (got.double() - truth).abs().max() <= 2 * (want.double() - truth).abs().max() + 1e-6


BQ_CASES = [(2, 8192, 1024, 0.1, 32), (2, 1024, 256, 0.2, 32), (3, 256, 64, 0.4, 32), (2, 64, 16, 0.8, 32),
            (1, 1000, 77, 0.3, 5), (1, 130, 9, 10.0, 70), (2, 50, 3, 1e-6, 4)]


@pytest.mark.parametrize('b,n,m,radius,u', BQ_CASES)
def test_ball_query(hip, oracle, gen, b, n, m, radius, u):
    pts = synth_cloud(gen, b, n, 's3dis')
    ctr = pts[:, :, torch.randperm(n, generator=gen)[:m]].contiguous()   # centres are cloud points (FPS output)
    want = oracle.ball_query(ctr, pts, radius, u)
    got = hip.ball_query(ctr.to(DEV), pts.to(DEV), radius, u)
    assert got.dtype == torch.int32 and torch.equal(got.cpu(), want)


@pytest.mark.parametrize('b,c,n,m,u', [(2, 32, 8192, 1024, 32), (2, 3, 8192, 1024, 32), (1, 67, 1000, 33, 7),
                                        (2, 259, 64, 16, 32), (1, 4, 50000, 10, 3), (1, 3, 300, 2000, 32)])
def test_grouping_fwd_bwd(hip, oracle, gen, b, c, n, m, u):
    f = torch.randn(b, c, n, generator=gen)
    idx = torch.randint(0, n, (b, m, u), generator=gen, dtype=torch.int32)
    idx[:, :, u // 2:] = idx[:, :, :1]    # ball_query-style padding: heavy duplicates in the backward
    assert torch.equal(hip.grouping_forward(f.to(DEV), idx.to(DEV)).cpu(), oracle.grouping_forward(f, idx))
    g = torch.randn(b, c, m, u, generator=gen)
    got, want = hip.grouping_backward(g.to(DEV), idx.to(DEV), n).cpu(), oracle.grouping_backward(g, idx, n)
    if n <= (1 << 20):   # kCsrMaxTargets
        assert torch.equal(got, want)
    else:
        assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('b,c,n,m', [(2, 3, 8192, 1024), (1, 5, 1001, 333), (4, 3, 1024, 512)])
def test_gather_fwd_bwd(hip, oracle, gen, b, c, n, m):
    f = torch.randn(b, c, n, generator=gen)
    idx = torch.randint(0, n, (b, m), generator=gen, dtype=torch.int32)
    assert torch.equal(hip.gather_features_forward(f.to(DEV), idx.to(DEV)).cpu(), oracle.gather_features_forward(f, idx))
    g = torch.randn(b, c, m, generator=gen)
    assert torch.equal(hip.gather_features_backward(g.to(DEV), idx.to(DEV), n).cpu(), oracle.gather_features_backward(g, idx, n))


FPS_CASES = [(2, 8192, 1024), (2, 1024, 256), (3, 256, 64), (2, 64, 16), (1, 1000, 100), (1, 5000, 50),
             (1, 17, 17), (1, 20000, 40), (1, 700, 1), (1, 3000, 64), (2, 12000, 64), (1, 16000, 32)]   # every launch shape of fps.hip


@pytest.mark.parametrize('b,n,m', FPS_CASES)
def test_fps(hip, oracle, gen, b, n, m):
    pts = synth_cloud(gen, b, n, 's3dis')      # includes exact duplicates -> exercises ties at distance 0
    assert torch.equal(hip.furthest_point_sampling(pts.to(DEV), m).cpu(), oracle.furthest_point_sampling(pts, m))


@pytest.mark.parametrize('side,m', [(12, 200), (8, 100), (20, 150), (6, 216)])
def test_fps_tie_rule_lattice(hip, oracle, side, m):
    # integer lattice: masses of exactly equidistant candidates -> the (k mod 512, k) rule decides every step.  N = 1728 and 8000:
    # ties between the lanes and waves of a 512-thread workgroup; N = 512 and 216: between the lanes of the one-wave kernel
    # (fps.hip: these steps leave the float-maximum path for the 64-bit key reduction)
    g = torch.arange(side, dtype=torch.float32)
    pts = torch.stack(torch.meshgrid(g, g, g, indexing='ij')).reshape(1, 3, -1).contiguous()
    assert torch.equal(hip.furthest_point_sampling(pts.to(DEV), m).cpu(), oracle.furthest_point_sampling(pts, m))


@pytest.mark.parametrize('b,c,m,n', [(2, 128, 1024, 8192), (2, 256, 64, 256), (2, 512, 16, 64), (1, 7, 2, 33),
                                      (1, 3, 1, 10)])
def test_three_nn_interpolate(hip, oracle, gen, b, c, m, n):
    pts = synth_cloud(gen, b, n, 's3dis')
    ctr = pts[:, :, torch.randperm(n, generator=gen)[:m]].contiguous()     # centres coincide with points (d = 0 clamp)
    feats = torch.randn(b, c, m, generator=gen)
    o_out, o_idx, o_w = oracle.three_nearest_neighbors_interpolate_forward(pts, ctr, feats)
    h_out, h_idx, h_w = hip.three_nearest_neighbors_interpolate_forward(pts.to(DEV), ctr.to(DEV), feats.to(DEV))
    assert torch.equal(h_idx.cpu(), o_idx)
    assert torch.equal(h_w.cpu(), o_w)
    assert torch.equal(h_out.cpu(), o_out)
    g = torch.randn(b, c, n, generator=gen)
    assert torch.equal(hip.three_nearest_neighbors_interpolate_backward(g.to(DEV), h_idx, h_w, m).cpu(),
                       oracle.three_nearest_neighbors_interpolate_backward(g, o_idx, o_w, m))


@pytest.mark.parametrize('shape', [(8, 64, 1024, 32), (2, 5, 33, 32), (1, 3, 7, 4), (2, 4, 9, 8), (3, 2, 5, 16), (1, 2, 3, 64)])
def test_neighbor_max_is_torch_max(hip, gen, shape):
    """csrc/pool.hip vs `x.max(dim=-1)` of torch on the same device (modules/pointnet.py:85 of the reference): values and winners
    bit-identical -- on ReLU outputs, i.e. with rows that are all zeros and rows with several equal maxima (first index wins) --
    and the backward equal to torch's fill + scatter."""
    from pvcnn_amd.modules.functional.pooling import neighbor_max
    x = torch.relu(torch.randn(*shape, generator=gen)).mul(4).round().div(4).to(DEV)        # coarse values: many ties
    x[0, 0, 0, :] = 0.0
    out, winners = hip.neighbor_max_forward(x)
    ref = x.max(dim=-1)
    assert torch.equal(out, ref.values)
    assert torch.equal(winners.long(), ref.indices)
    g = torch.randn(shape[:-1], generator=gen).to(DEV)
    want = torch.zeros_like(x).scatter_(-1, ref.indices.unsqueeze(-1), g.unsqueeze(-1))
    assert torch.equal(hip.neighbor_max_backward(g, winners, shape[-1]), want)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    (neighbor_max(xa) * g).sum().backward()
    (xb.max(dim=-1).values * g).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    x[-1, -1, -1, 1] = float('nan')                                                         # a NaN wins, like torch.max
    assert torch.isnan(hip.neighbor_max_forward(x)[0][-1, -1, -1]) and hip.neighbor_max_forward(x)[1][-1, -1, -1].item() == 1
    odd = torch.randn(2, 3, 5, 12, generator=gen).to(DEV)                                   # K = 12: not covered -> torch.max itself
    assert not hip.neighbor_max_supported(12) and torch.equal(neighbor_max(odd), odd.max(dim=-1).values)


@pytest.mark.parametrize('shape', [(16, 1024, 4096), (2, 3, 100), (1, 5, 8), (2, 2, 4100), (3, 1, 2048), (1, 2, 4)])
def test_row_argmax_is_torch_max(hip, gen, shape):
    """csrc/pool.hip row_argmax_kernel (the global max-pool over the points, models/s3dis/pvcnn.py:41-43) vs `x.max(dim=-1)`:
    the same winners and values on ReLU outputs (all-zero rows, repeated maxima: the first index), and a NaN wins."""
    x = torch.relu(torch.randn(*shape, generator=gen)).mul(2).round().div(2).to(DEV)
    x[0, 0, :] = 0.0
    winners, values = hip.row_argmax(x, with_values=True)
    ref = x.max(dim=-1)
    assert torch.equal(winners, ref.indices) and torch.equal(values, ref.values)
    if shape[-1] > 5:
        x[-1, -1, 5] = float('nan')
        w2, v2 = hip.row_argmax(x, with_values=True)
        assert w2[-1, -1].item() == 5 and torch.isnan(v2[-1, -1])


def test_empty_inputs(hip):
    z = torch.zeros
    assert hip.avg_voxelize_forward(z(0, 4, 16, device=DEV), z(0, 3, 16, dtype=torch.int32, device=DEV), 4)[0].shape == (0, 4, 64)
    out, ind, cnt = hip.avg_voxelize_forward(z(2, 4, 0, device=DEV), z(2, 3, 0, dtype=torch.int32, device=DEV), 4)
    assert out.abs().sum().item() == 0 and cnt.sum().item() == 0     # no points: empty grid, fully written
    assert hip.ball_query(z(1, 3, 0, device=DEV), z(1, 3, 10, device=DEV), 1.0, 4).shape == (1, 0, 4)
    assert hip.grouping_forward(z(1, 2, 5, device=DEV), z(1, 0, 3, dtype=torch.int32, device=DEV)).shape == (1, 2, 0, 3)
    gx = hip.grouping_backward(z(1, 2, 0, 3, device=DEV), z(1, 0, 3, dtype=torch.int32, device=DEV), 5)
    assert gx.shape == (1, 2, 5) and gx.abs().sum().item() == 0


def test_input_contract_errors(hip):
    f = torch.rand(1, 2, 8, device=DEV)
    with pytest.raises(RuntimeError, match='int tensor'):
        hip.avg_voxelize_forward(f, torch.zeros(1, 3, 8, dtype=torch.int64, device=DEV), 2)
    with pytest.raises(RuntimeError, match='float tensor'):
        hip.avg_voxelize_forward(f.double(), torch.zeros(1, 3, 8, dtype=torch.int32, device=DEV), 2)
    with pytest.raises(RuntimeError, match='contiguous'):
        hip.grouping_forward(torch.rand(1, 8, 2, device=DEV).transpose(1, 2), torch.zeros(1, 2, 2, dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError):
        hip.trilinear_devoxelize_forward(4, True, torch.rand(1, 3, 8, device=DEV), torch.rand(1, 2, 60, device=DEV))   # 60 != 4^3


def test_runs_on_the_current_stream(hip, oracle, gen):
    # work is enqueued on torch's current stream (no legacy-default-stream launches like vox.cu:114)
    feat, vox, norm = _vox_inputs(gen, 2, 16, 2048, 16)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f_d, v_d = feat.to(DEV, non_blocking=True), vox.to(DEV, non_blocking=True)
        out = hip.avg_voxelize_forward(f_d, v_d, 16)[0]
        dev = hip.trilinear_devoxelize_forward(16, False, norm.to(DEV), out)[0]
    s.synchronize()
    o_out = oracle.avg_voxelize_forward(feat, vox, 16)[0]
    assert torch.equal(out.cpu(), o_out)
    assert torch.equal(dev.cpu(), oracle.trilinear_devoxelize_forward(16, False, norm, o_out)[0])
