"""voxel_coords / norm_coords are BIT-IDENTICAL to the reference formulation on the same device.

Reference: modules/voxelization.py:16-25 -- `coords.mean(2)`, `norm(dim=1).max(dim=2)`, then elementwise
centre / divide / +0.5 / *R / clamp / round.  north_star demands bit-exact voxel_coords.  The default GPU path of
pvcnn_amd.modules.Voxelization calls the reference's two torch reductions and fuses everything behind them into
one kernel (csrc/voxelize.hip: voxel_coords_tail_kernel); here its outputs are compared with torch.equal -- no
boundary mask, every point -- against `Voxelization.normalized_coords`, which is the reference formulation op
by op, on the inputs of every BASELINE config (workload.make_*).  The opt-in single-launch kernel
(`Voxelization.single_launch`, order-free statistics) is compared too and its mismatch COUNT is printed: it is
allowed to differ where a coordinate sits on a rounding boundary, which is why it is not the default.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _coords(cfg):
    from pvcnn_amd import workload
    if cfg == 'cfg2':      # PVCNN S3DIS B=16 N=4096: coords are a channel SLICE of the (B,9,N) input
        x, _ = workload.make_s3dis_batch(16, 4096, device=DEV)
        return x[:, :3, :]
    if cfg == 'cfg3':      # PVCNN++ S3DIS B=8 N=8192 (contiguous copy, models/s3dis/pvcnnpp.py:45)
        x, _ = workload.make_s3dis_batch(8, 8192, device=DEV)
        return x[:, :3, :].contiguous()
    if cfg == 'cfg3-deep':  # a deeper PVCNN++ level: 1024 FPS-like centres
        x, _ = workload.make_s3dis_batch(8, 8192, device=DEV)
        return x[:, :3, ::8].contiguous()
    if cfg == 'cfg4':      # ShapeNet B=64 N=2048, normalize=False, slice of a (B,22,N) input
        x, _ = workload.make_shapenet_batch(64, 2048, device=DEV)
        return x[:, :6, :][:, :3, :]
    if cfg == 'cfg5':      # Frustum B=32 N=1024, slice of (B,4,N)
        x, _ = workload.make_frustum_batch(32, 1024, device=DEV)
        return x['features'][:, :3, :]
    if cfg == 'ragged':    # N not a multiple of 4 (scalar kernel), odd batch stride
        x, _ = workload.make_s3dis_batch(3, 1001, device=DEV)
        return x[:, :3, :]
    raise KeyError(cfg)


CASES = [('cfg2', 32, True, 0), ('cfg2', 16, True, 0), ('cfg3', 32, True, 0), ('cfg3', 16, True, 0), ('cfg3', 8, True, 0),
         ('cfg3-deep', 16, True, 0), ('cfg4', 32, False, 0), ('cfg4', 16, False, 0), ('cfg5', 16, True, 0), ('cfg5', 12, True, 0),
         ('cfg5', 12, True, 1e-15), ('ragged', 12, True, 0), ('ragged', 5, False, 0)]


@pytest.mark.parametrize('cfg,r,normalize,eps', CASES)
def test_voxel_coords_bit_identical_to_the_reference_formulation(hip, cfg, r, normalize, eps):
    from pvcnn_amd.modules import Voxelization
    coords = _coords(cfg)
    vox_mod = Voxelization(r, normalize=normalize, eps=eps)
    want_norm = vox_mod.normalized_coords(coords)                     # the reference's ops, same device
    want_vox = torch.round(want_norm).to(torch.int32)
    got_norm, got_vox = vox_mod.grid_coordinates(coords)
    assert got_norm.dtype == torch.float32 and got_vox.dtype == torch.int32
    assert got_norm.is_contiguous() and got_vox.is_contiguous()
    assert torch.equal(got_norm, want_norm), f'{(got_norm != want_norm).sum().item()} norm_coords differ'
    assert torch.equal(got_vox, want_vox), f'{(got_vox != want_vox).sum().item()} voxel ids differ'
    assert 0 <= int(got_vox.min()) and int(got_vox.max()) <= r - 1

    # the opt-in one-launch kernel: report how often it disagrees (it must still be within an ulp-scale distance)
    alt_norm, alt_vox = hip.voxel_coords(coords.contiguous(), r, normalize, eps)
    n_norm, n_vox = (alt_norm != want_norm).sum().item(), (alt_vox != want_vox).sum().item()
    print(f'[single-launch opt-in] {cfg} R={r} normalize={normalize}: {n_norm} of {want_norm.numel()} norm_coords and '
          f'{n_vox} voxel ids differ from the torch formulation')
    assert (alt_norm - want_norm).abs().max().item() <= 1e-5 * r


def test_module_forward_uses_the_bit_identical_prepass_and_shares_it(hip):
    """Voxelization.forward: same grid as avg_voxelize on the reference formulation's voxel ids; layers that see the
    same coords tensor and R share ONE pre-pass (the memo returns the very same tensors)."""
    from pvcnn_amd.modules import Voxelization
    from pvcnn_amd.modules import functional as F
    coords = _coords('cfg2')
    feats = torch.randn(16, 8, 4096, device=DEV)
    a, b = Voxelization(16), Voxelization(16)
    grid_a, norm_a = a(feats, coords)
    grid_b, norm_b = b(feats, coords)
    assert norm_a is norm_b, 'second layer with the same (coords, R) must reuse the first pre-pass'
    want_norm = a.normalized_coords(coords)
    assert torch.equal(norm_a, want_norm)
    assert torch.equal(grid_a, F.avg_voxelize(feats, torch.round(want_norm).to(torch.int32), 16))
    assert torch.equal(grid_a, grid_b)
    # an in-place change of the coordinates invalidates the memo
    base = coords.clone()
    n1 = Voxelization(16)(feats, base)[1]
    base.mul_(0.5).add_(0.1)
    n2 = Voxelization(16)(feats, base)[1]
    assert torch.equal(n2, a.normalized_coords(base)) and n1 is not n2
