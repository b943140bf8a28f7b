"""The RANGE CONTRACT of the default "f16x2" arithmetic (include/pvcnn_hip.h, "amax buffers"; csrc/split16.h).

Both fp32 operands of a product are split into fp16 hi + lo after scaling by a power of two.  fp16 has 5 exponent bits, so an
operand keeps all 22 bits only within 2^-17 of the magnitude the scale was chosen for; below that it loses low bits gradually
(absolute error <= 2^-38 of that magnitude per product).  Round 2 chose ONE scale per activation / gradient tensor: a single
outlier, or a gradient with a heavy tail, then cost every ordinary element its low bits -- invisible to a metric normalised by
the output's largest entry.  Since round 3 the forward / backward-data kernels scale per WORKGROUP TILE (Conv3d: the z rows of the
halo tile; 1x1 GEMM: 256 points), from an amax buffer that costs the same single read of the tensor as the old global maximum.

What is asserted here, element by element against fp64 and relative to each output element's OWN magnitude scale (the sum of
|products| feeding it: the quantity fp32 accumulation itself is accurate to):
  * an outlier of 1e8 x the typical magnitude leaves every output whose tile does not contain it at the ordinary fp32-class error;
  * a tensor whose magnitude decays smoothly over 12 decades across positions (a sparse / heavy-tailed gradient) is fp32-class
    EVERYWHERE, including where it is 1e-12 of the maximum -- the single-scale mode (amax_seg = 0, kept selectable) flushes that
    region to zero, which the same test demonstrates;
  * channels 1e7 apart in magnitude: the small channels' own contribution is carried to fp32-class accuracy relative to the
    largest term of the dot product it enters (what fp32 accumulation preserves), bounded by the documented 2^-38 * tile max;
  * the backward-weight kernels (one global scale: they REDUCE over positions) stay within 1e-5 of fp64 on the same tensors.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
FP32_CLASS = 2e-6          # |err| / sum |products|: a few fp32 roundings


def _conv_ref(x, w):
    """fp64 convolution and the per-element magnitude scale sum |w| * |x| over the receptive field."""
    ref = F.conv3d(x.double(), w.double(), padding=1)
    mag = F.conv3d(x.double().abs(), w.double().abs(), padding=1)
    return ref, mag


def _pw_ref(x, w):
    return torch.einsum('mk,bkn->bmn', w.double(), x.double()), torch.einsum('mk,bkn->bmn', w.double().abs(), x.double().abs())


def test_amax_buffer_layout(hip):
    """absmax_tiles == per-segment max over channels (bit patterns), [0] == the global maximum; ragged sizes included."""
    g = torch.Generator().manual_seed(1)
    for b, c, n, seg in [(2, 5, 1000, 256), (3, 7, 32 ** 3, 32), (1, 1, 12 ** 3, 12), (2, 64, 16 ** 3, 16), (2, 3, 777, 256), (1, 9, 5, 4)]:
        x = torch.randn(b, c, n, generator=g).to(DEV)
        x[0, 0, n // 2] = -123.5
        got = hip.absmax_tiles(x, seg)
        nseg = (n + seg - 1) // seg
        pad = torch.zeros(b, c, nseg * seg, device=DEV)
        pad[:, :, :n] = x.abs()
        want = pad.view(b, c, nseg, seg).amax(dim=(1, 3)).reshape(-1)
        assert got.numel() == 1 + b * nseg
        assert torch.equal(got[1:], want.view(torch.int32)), (b, c, n, seg)
        assert got[0].item() == torch.tensor(123.5).view(torch.int32).item()
    assert hip.absmax_tiles(torch.zeros(2, 3, 64, device=DEV), 16).abs().max().item() == 0


@pytest.mark.parametrize('r', [16, 32])
def test_conv3d_outlier_only_costs_its_own_tiles(hip, r):
    g = torch.Generator().manual_seed(2)
    b, ci, co = 2, 64, 64
    x = torch.randn(b, ci, r, r, r, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ox, oy, oz = r // 2, r // 2, 5
    x[1, 3, ox, oy, oz] = 1.0e8
    ref, mag = _conv_ref(x, w)
    seen = {}
    for mode, amax in (('tiles', hip.conv_amax(x)), ('single scale', hip.absmax_bits(x))):
        y = hip.conv3d_forward_split(x, w, None, 2, amax=amax)
        rel = (y.double() - ref).abs() / mag
        # "away from the outlier": every voxel of cloud 0, and of cloud 1 further than a workgroup tile + halo (8 voxels in x / y)
        away = torch.ones_like(rel, dtype=torch.bool)
        away[1, :, max(ox - 8, 0):ox + 9, max(oy - 8, 0):oy + 9, :] = False
        worst = rel[away].max().item()
        print(f'[range] conv R={r} 1e8 outlier, {mode}: worst error away from it {worst:.2e} (near it {rel[~away].max().item():.2e})')
        seen[mode] = worst
    assert seen['tiles'] < FP32_CLASS, seen
    # the cliff this round closed: under one global scale every ordinary element carried ~12 bits (quantum 2^-38 of the outlier)
    assert seen['single scale'] > 10 * seen['tiles'], seen


def test_conv3d_smoothly_decaying_gradient_is_fp32_class_everywhere(hip):
    g = torch.Generator().manual_seed(3)
    b, c, r = 2, 64, 32
    x = torch.randn(b, c, r, r, r, generator=g)
    decay = torch.logspace(0, -12, r).view(1, 1, r, 1, 1)          # 12 decades along x: half a decade per plane
    x = (x * decay).to(DEV)
    w = (torch.randn(c, c, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ref, mag = _conv_ref(x, w)
    y = hip.conv3d_backward_data_split(x, w, 2)                     # the gradient path: grad_x = conv(grad_y, w')
    refb = F.conv_transpose3d(x.double(), w.double(), padding=1)
    magb = F.conv_transpose3d(x.double().abs(), w.double().abs(), padding=1)
    rel = (y.double() - refb).abs() / magb
    print(f'[range] conv backward-data on a 12-decade gradient: worst per-element error {rel.max().item():.2e}; '
          f'in the 1e-12 planes {rel[:, :, -2:].max().item():.2e}')
    assert rel.max().item() < FP32_CLASS
    y1 = hip.conv3d_backward_data_split(x, w, 2, amax=hip.absmax_bits(x))
    rel1 = (y1.double() - refb).abs() / magb
    assert rel1[:, :, -2:].max().item() > 1e-2                      # single scale: those planes flush to (nearly) nothing
    # backward-weight reduces over positions: one global scale is the right one, 1e-5 of fp64 as before
    gy = (torch.randn(b, c, r, r, r, generator=g) * decay).to(DEV)
    xin = torch.randn(b, c, r, r, r, generator=g).to(DEV)
    wd = torch.zeros(c, c, 3, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    F.conv3d(xin.double(), wd, padding=1).backward(gy.double())
    gw = hip.conv3d_backward_weight_f16(xin, gy, hip.conv_amax(xin), hip.conv_amax(gy))
    err = ((gw.double() - wd.grad).abs().max() / wd.grad.abs().max()).item()
    print(f'[range] conv backward-weight with a 12-decade grad_y: {err:.2e} of the largest entry')
    assert err < 1e-5


def test_pointwise_outlier_and_decay(hip):
    g = torch.Generator().manual_seed(4)
    b, k, m, n = 2, 256, 512, 4096
    w = (torch.randn(m, k, generator=g) * 0.05).to(DEV)
    x = torch.randn(b, k, n, generator=g).to(DEV)
    x[1, 7, 1000] = 1.0e8
    ref, mag = _pw_ref(x, w)
    y = hip.pwconv_forward_split(x, w, None, 2)
    rel = (y.double() - ref).abs() / mag
    away = torch.ones_like(rel, dtype=torch.bool)
    away[1, :, 768:1024] = False                                    # the outlier's 256-point tile
    print(f'[range] 1x1 GEMM 1e8 outlier: worst error outside its tile {rel[away].max().item():.2e}, inside {rel[~away].max().item():.2e}')
    assert rel[away].max().item() < FP32_CLASS
    y1 = hip.pwconv_forward_split(x, w, None, 2, amax=hip.absmax_bits(x))
    assert ((y1.double() - ref).abs() / mag)[away].max().item() > 10 * rel[away].max().item()      # single scale: the old cliff
    # log-uniform over 12 decades along the points (sparse-gradient shape), backward-data
    gy = (torch.randn(b, m, n, generator=g) * torch.logspace(0, -12, n).view(1, 1, n)).to(DEV)
    refb = torch.einsum('mk,bmn->bkn', w.double(), gy.double())
    magb = torch.einsum('mk,bmn->bkn', w.double().abs(), gy.double().abs())
    gx = hip.pwconv_backward_data_split(gy, w, 2)
    relb = (gx.double() - refb).abs() / magb
    print(f'[range] 1x1 backward-data on a 12-decade gradient: worst per-element error {relb.max().item():.2e}')
    assert relb.max().item() < FP32_CLASS
    gw = hip.pwconv_backward_weight_f16(x, gy, hip.pw_amax(x), hip.pw_amax(gy))
    gwr = torch.einsum('bmn,bkn->mk', gy.double(), x.double())
    assert ((gw.double() - gwr).abs().max() / gwr.abs().max()).item() < 1e-5


def test_channels_seven_decades_apart(hip):
    """Per-channel magnitudes 1e7 apart in one tensor: an output that mixes them is accurate relative to its largest term (what
    fp32 accumulation preserves); an output whose weights on the large channels are EXACTLY zero sees only the small channels --
    there the documented bound applies: |err| <= 2^-38 * tile max * sum |w| per product (with the 22 bits the small operands keep
    when they are within 2^-17 of the tile maximum; here they are 2^-23 below it: 16 bits)."""
    g = torch.Generator().manual_seed(5)
    b, k, m, n = 2, 64, 128, 2048
    x = torch.randn(b, k, n, generator=g)
    x[:, :8] *= 1.0e7
    x = x.to(DEV)
    w = (torch.randn(m, k, generator=g) * 0.1)
    w[:16, :8] = 0.0                                                 # outputs 0..15 see the small channels only
    w = w.to(DEV)
    ref, mag = _pw_ref(x, w)
    y = hip.pwconv_forward_split(x, w, None, 2)
    err = (y.double() - ref).abs()
    mixed = (err / mag)[:, 16:].max().item()
    small_only = (err / mag)[:, :16].max().item()
    bound = 2.0 ** -38 * x.abs().max().item() * w.abs().sum(dim=1).max().item() * 2   # documented absolute bound (x2: lo*lo' term, fp32 adds)
    print(f'[range] channels 1e7 apart: mixed outputs {mixed:.2e} of their magnitude; small-channel-only outputs {small_only:.2e} '
          f'(absolute {err[:, :16].max().item():.2e} <= documented bound {bound:.2e})')
    assert mixed < FP32_CLASS
    assert err[:, :16].max().item() <= bound


@pytest.mark.parametrize('shape,seg', [((16, 64, 32, 32, 32), 32), ((3, 24, 16, 16, 16), 16), ((2, 10, 12, 12, 12), 12),
                                       ((4, 64, 4096), 256), ((2, 7, 1001), 256), ((3, 130, 2, 600), 256),
                                       ((2, 512, 1024), 256)])    # (the S <= 4096 shapes split their channels over 2-4 workgroups)
def test_batchnorm_apply_passes_emit_the_buffer_and_the_same_bits(hip, shape, seg):
    """The position-block-major apply passes (csrc/bnact.hip) write bit for bit what the classic passes write, forward and backward
    (batch-strided gradient included), and the amax buffer they emit equals absmax_tiles of that output."""
    g = torch.Generator().manual_seed(6)
    b, c = shape[:2]
    x = torch.randn(*shape, generator=g).to(DEV).view(b, c, -1)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    wide = torch.randn(b, c + 5, x.shape[2], generator=g).to(DEV) * 1e-3
    gy = wide[:, 2:2 + c]                                            # a channel slice: clouds further apart than C * S
    for training in (True, False):
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        y0, mean, rstd = hip.bnact_forward(x, gamma, beta, rm.clone(), rv.clone(), training, 0.1, 1e-4, 0.1)
        y1, mean1, rstd1, amax = hip.bnact_forward(x, gamma, beta, rm.clone(), rv.clone(), training, 0.1, 1e-4, 0.1, amax_seg=seg)
        assert torch.equal(y0, y1) and torch.equal(mean, mean1) and torch.equal(rstd, rstd1)
        assert torch.equal(amax, hip.absmax_tiles(y1, seg))
        g0 = hip.bnact_backward(x, gy, gamma, beta, mean, rstd, 0.1, training)
        g1 = hip.bnact_backward(x, gy, gamma, beta, mean, rstd, 0.1, training, amax_seg=seg)
        for a, b_ in zip(g0, g1[:3]):
            assert torch.equal(a, b_)
        assert torch.equal(g1[3], hip.absmax_tiles(g1[0], seg))
