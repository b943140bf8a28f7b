"""PVConv's FIRST BatchNorm3d + LeakyReLU folded into the SECOND convolution's staging (forward and
backward-weight; csrc/conv3d_bf16.hip / conv3d_wgrad_f16.hip, XF), and max |grad| emitted by the BatchNorm backward's apply
pass instead of a separate absmax pass (csrc/bnact.hip).  The consumer applies bnact_apply_kernel's very expressions, so
everything here is compared BIT FOR BIT with the unfused ops (reference composition: modules/pvconv.py:20-27)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _bn_of(c, gen, affine=True):
    mean = torch.randn(c, generator=gen) * 0.3
    rstd = torch.rand(c, generator=gen) * 1.5 + 0.3
    gamma = (torch.rand(c, generator=gen) + 0.5) * torch.where(torch.rand(c, generator=gen) < 0.2, -1.0, 1.0) if affine else None
    beta = torch.randn(c, generator=gen) * 0.4 if affine else None
    dev = lambda t: t.to(DEV) if t is not None else None
    return dev(gamma), dev(beta), dev(mean), dev(rstd), 0.1


def _materialise(hip, x, bn):
    """act(bn(x)) written out by the stand-alone BatchNorm + LeakyReLU pass with the same statistics."""
    gamma, beta, mean, rstd, slope = bn
    b, c = x.shape[:2]
    y, _, _ = hip.bnact_forward(x.view(b, c, -1), gamma, beta, None, None, False, 0.0, 0.0, slope, stats=(mean, rstd))
    return y.view(x.shape)


@pytest.mark.parametrize('shape', [(2, 64, 16), (1, 64, 32), (2, 20, 16), (3, 9, 12), (2, 24, 8), (1, 128, 16), (2, 3, 5)])
@pytest.mark.parametrize('affine', [True, False])
def test_absmax_through_the_transform(hip, gen, shape, affine):
    b, c, r = shape
    x = torch.randn(b, c, r, r, r, generator=gen).to(DEV) * 3
    bn = _bn_of(c, gen, affine)
    got = hip.bnact_absmax_bits(x, bn)
    want = hip.absmax_bits(_materialise(hip, x, bn))
    assert torch.equal(got, want)


@pytest.mark.parametrize('b,ci,co,r', [(2, 64, 64, 16),      # (4,4,16) / (2,4,16) vector tiles
                                       (16, 64, 64, 16),     # enough tiles for the (4,4,16) choice
                                       (1, 64, 64, 32),      # (2,4,32)
                                       (8, 32, 64, 32),      # 512-voxel tile (4,4,32)
                                       (2, 20, 40, 16),      # channel counts that are not multiples of 16 / 64
                                       (2, 16, 16, 12),      # vector staging with a ragged z row (R = 12 < tz = 16)
                                       (2, 16, 32, 10),      # scalar staging (R % 4 != 0): (4,4,16,false)
                                       (1, 8, 8, 6),         # scalar staging, (4,8,8,false)
                                       (2, 24, 24, 8)])      # (4,8,8)
@pytest.mark.parametrize('affine', [True, False])
def test_folded_forward_is_bit_identical(hip, gen, b, ci, co, r, affine):
    x = torch.randn(b, ci, r, r, r, generator=gen).to(DEV) * 2 + 0.3
    w = (torch.randn(co, ci, 3, 3, 3, generator=gen) * 0.05).to(DEV)
    bias = torch.randn(co, generator=gen).to(DEV)
    bn = _bn_of(ci, gen, affine)
    act = _materialise(hip, x, bn)
    want, want_part = hip.conv3d_forward_split(act, w, bias, 2, want_stats=True)
    got, got_part = hip.conv3d_forward_split_bnact(x, w, bias, bn, want_stats=True)
    assert torch.equal(got, want)
    assert torch.equal(got_part, want_part)
    # ... and against fp64 on the materialised activation (the f16x2 bar of tests/test_gpu_conv3d.py)
    ref = torch.nn.functional.conv3d(act.double(), w.double(), bias.double(), padding=1)
    assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize('b,ci,co,r', [(2, 64, 64, 16), (1, 64, 64, 32), (2, 20, 40, 16), (1, 128, 64, 16), (2, 9, 16, 32)])
@pytest.mark.parametrize('affine', [True, False])
def test_folded_backward_weight_is_bit_identical(hip, gen, b, ci, co, r, affine):
    x = torch.randn(b, ci, r, r, r, generator=gen).to(DEV) * 2 - 0.2
    gy = torch.randn(b, co, r, r, r, generator=gen).to(DEV) * 0.01
    bn = _bn_of(ci, gen, affine)
    act = _materialise(hip, x, bn)
    want_w, want_b = hip.conv3d_backward_weight_f16(act, gy, with_bias=True)
    got_w, got_b = hip.conv3d_backward_weight_f16_bnact(x, gy, None, None, bn, with_bias=True)
    assert torch.equal(got_w, want_w) and torch.equal(got_b, want_b)


@pytest.mark.parametrize('shape', [(2, 64, 4096), (3, 24, 1000), (2, 7, 1001), (16, 64, 16 ** 3)])
@pytest.mark.parametrize('training', [True, False])
def test_batchnorm_backward_emits_the_gradient_maximum(hip, gen, shape, training):
    b, c, s = shape
    x = torch.randn(b, c, s, generator=gen).to(DEV)
    g = torch.randn(b, c, s, generator=gen).to(DEV) * 1e-3
    gamma, beta, mean, rstd, slope = _bn_of(c, gen)
    plain = hip.bnact_backward(x, g, gamma, beta, mean, rstd, slope, training)
    gx, gg, gb, amax = hip.bnact_backward(x, g, gamma, beta, mean, rstd, slope, training, want_amax=True)
    for a, p in zip((gx, gg, gb), plain):
        assert torch.equal(a, p)
    assert torch.equal(amax, hip.absmax_bits(gx))
    # a second call re-arms the word (it is zeroed on the device, by the finalize launch)
    gx2, _, _, amax2 = hip.bnact_backward(x, g * 0.5, gamma, beta, mean, rstd, slope, training, want_amax=True)
    assert torch.equal(amax2, hip.absmax_bits(gx2)) and not torch.equal(amax2, amax)


@pytest.mark.parametrize('r,cin,cout,with_se', [(16, 9, 32, False), (32, 16, 16, False), (16, 64, 64, True)])
def test_pvconv_with_the_fold_is_bit_identical_to_without(hip, r, cin, cout, with_se):
    """Whole PVConv, train mode (batch statistics from the first convolution's epilogue) and eval mode (running statistics):
    output, input gradient, every parameter gradient and every buffer agree bit for bit with the fold switched off."""
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd.modules.functional._autograd import native
    torch.manual_seed(23)
    be = native()
    assert be.conv_math == 'f16x2'
    default = type(be).has_conv3d_bnact_fold                 # the fold is opt-in (PVCNN_FOLD_BN=1): switched on here
    folded = PVConv(cin, cout, 3, r, with_se=with_se).to(DEV).train()
    plain = copy.deepcopy(folded)
    feats = torch.randn(2, cin, 1500, device=DEV)
    coords = torch.rand(2, 3, 1500, device=DEV) * 2 - 1
    fa, fb = feats.clone().requires_grad_(), feats.clone().requires_grad_()
    type(be).has_conv3d_bnact_fold = True
    try:
        ya, _ = folded((fa, coords))
    finally:
        type(be).has_conv3d_bnact_fold = default
    # the folded node is in the graph: walk it
    used, seen, stack = [], set(), [ya.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        used.append(type(fn).__name__)
        stack.extend(f for f, _ in fn.next_functions)
    assert any('BnActVoxelConv3d' in n for n in used), used
    ya.square().sum().backward()
    try:
        type(be).has_conv3d_bnact_fold = False
        yb, _ = plain((fb, coords))
        yb.square().sum().backward()
    finally:
        type(be).has_conv3d_bnact_fold = default
    assert torch.equal(ya, yb)
    assert torch.equal(fa.grad, fb.grad)
    for (na, pa), (_, pb) in zip(folded.named_parameters(), plain.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), na
    for (na, ba), (_, bb) in zip(folded.named_buffers(), plain.named_buffers()):
        assert torch.equal(ba, bb), na
    folded.eval(); plain.eval()
    with torch.no_grad():
        try:
            type(be).has_conv3d_bnact_fold = True
            ya, _ = folded((feats, coords))
            type(be).has_conv3d_bnact_fold = False
            yb, _ = plain((feats, coords))
        finally:
            type(be).has_conv3d_bnact_fold = default
    assert torch.equal(ya, yb)


def test_gradient_maxima_are_handed_over_not_measured_again(hip):
    """One PVConv forward + backward: with the BatchNorm backward emitting max |grad| the only absmax passes left are the
    ones over tensors no BatchNorm backward wrote (the voxelized input of the first convolution); results are unchanged."""
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd.modules.functional._autograd import native
    torch.manual_seed(29)
    be = native()
    layer = PVConv(16, 32, 3, 16).to(DEV).train()
    twin = copy.deepcopy(layer)
    feats = torch.randn(2, 16, 2048, device=DEV)
    coords = torch.rand(2, 3, 2048, device=DEV)

    def run(mod, emit):
        calls = []
        orig = type(be).absmax_bits
        type(be).absmax_bits = lambda self, t: (calls.append(tuple(t.shape)), orig(self, t))[1]
        prev, prev_fold = type(be).has_bnact_bwd_absmax, type(be).has_conv3d_bnact_fold
        type(be).has_bnact_bwd_absmax = emit                   # both are opt-in (PVCNN_BWD_AMAX=1, PVCNN_FOLD_BN=1): switched on here
        type(be).has_conv3d_bnact_fold = True
        try:
            f = feats.clone().requires_grad_()
            y, _ = mod((f, coords))
            y.square().sum().backward()
        finally:
            type(be).absmax_bits = orig
            type(be).has_bnact_bwd_absmax, type(be).has_conv3d_bnact_fold = prev, prev_fold
        return y, f.grad, calls

    ya, ga, calls_a = run(layer, True)
    yb, gb, calls_b = run(twin, False)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    for (n, pa), (_, pb) in zip(layer.named_parameters(), twin.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), n
    # without the hand-over: conv1 input, grad of conv2 output, grad of conv1 output (+ nothing for the folded activation);
    # with it: only the first convolution's input
    assert len(calls_b) == 3, calls_b
    assert len(calls_a) == 1, calls_a
