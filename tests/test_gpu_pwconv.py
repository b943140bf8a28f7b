"""1x1 convolutions of SharedMLP on the MFMA GEMM kernels (csrc/pointwise.hip) vs an fp64 evaluation.

Reference call sites: modules/shared_mlp.py:9-25 (nn.Conv1d / nn.Conv2d, kernel 1).  Tolerance: fp32
round-off of a K-term dot product, 1e-5 relative to the largest magnitude of the result."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

SHAPES = [  # (B, Ci, Co, spatial)
    (2, 9, 64, (4096,)),        # PVConv point branch, first layer (fast path)
    (2, 64, 13, (1000,)),       # Co not a multiple of 4: scalar staging
    (1, 5, 7, (37,)),           # everything ragged
    (1, 200, 260, (132,)),      # several K chunks, Co and Ci beyond one tile, N tail
    (2, 35, 64, (16, 32)),      # Conv2d after grouping: (B, C+3, M, U)
    (3, 128, 128, (256,)),
]


def _rel(a, b):
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize('b,ci,co,sp', SHAPES)
def test_pointwise_conv_forward_backward(hip, b, ci, co, sp):
    from pvcnn_amd.modules.functional.pwconv import pointwise_conv
    torch.manual_seed(ci * 131 + co)
    dev = 'cuda:0'
    x = torch.randn(b, ci, *sp, device=dev, requires_grad=True)
    w = (torch.randn(co, ci, *([1] * len(sp)), device=dev) * 0.1).requires_grad_()
    bias = torch.randn(co, device=dev, requires_grad=True)
    gy = torch.randn(b, co, *sp, device=dev)
    y = pointwise_conv(x, w, bias)
    y.backward(gy)
    xd, wd, bd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_(), bias.detach().double().requires_grad_()
    conv = F.conv1d if len(sp) == 1 else F.conv2d
    yd = conv(xd, wd, bd)
    yd.backward(gy.double())
    assert y.shape == yd.shape
    assert _rel(y.detach(), yd.detach()) < 1e-5
    assert _rel(x.grad, xd.grad) < 1e-5
    assert _rel(w.grad, wd.grad) < 1e-5
    assert _rel(bias.grad, bd.grad) < 1e-5


def test_pointwise_conv_without_bias_and_partial_grads(hip):
    from pvcnn_amd.modules.functional.pwconv import pointwise_conv
    dev = 'cuda:0'
    x = torch.randn(2, 16, 64, device=dev)                      # no grad wrt x
    w = torch.randn(8, 16, 1, device=dev, requires_grad=True)
    y = pointwise_conv(x, w, None)
    y.sum().backward()
    ref = F.conv1d(x.double(), w.detach().double())
    assert _rel(y.detach(), ref) < 1e-5
    assert _rel(w.grad, x.double().sum(dim=(0, 2)).view(1, 16, 1).expand(8, 16, 1)) < 1e-5


def test_shared_mlp_uses_the_native_gemm_and_matches_torch(hip):
    """SharedMLP through run_layers (own GEMM + fused BN/ReLU) vs the same modules through torch."""
    from pvcnn_amd.modules import SharedMLP
    torch.manual_seed(3)
    dev = 'cuda:0'
    mine = SharedMLP(32, [64, 48]).to(dev).train()
    x = torch.randn(4, 32, 512, device=dev, requires_grad=True)
    y = mine(x)
    y.square().mean().backward()
    g_mine = [p.grad.clone() for p in mine.parameters()]
    gx_mine = x.grad.clone()
    # plain torch evaluation of the very same layer stack
    for p in mine.parameters():
        p.grad = None
    for m in mine.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.reset_running_stats()
    x2 = x.detach().clone().requires_grad_()
    y2 = nn.Sequential.forward(mine.layers, x2)
    y2.square().mean().backward()
    assert torch.allclose(y, y2, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gx_mine, x2.grad, rtol=1e-3, atol=1e-6)
    for a, p in zip(g_mine, mine.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-3, atol=1e-5)


def test_pointwise_conv_is_deterministic(hip):
    from pvcnn_amd.modules.functional.pwconv import pointwise_conv
    dev = 'cuda:0'
    torch.manual_seed(0)
    x = torch.randn(4, 96, 2048, device=dev, requires_grad=True)
    w = torch.randn(160, 96, 1, device=dev, requires_grad=True)
    bias = torch.randn(160, device=dev, requires_grad=True)
    outs = []
    for _ in range(2):
        for t in (x, w, bias):
            t.grad = None
        y = pointwise_conv(x, w, bias)
        y.backward(torch.ones_like(y))
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), bias.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('b,ci,co,n', [(2, 9, 64, 4096), (3, 64, 128, 1000), (1, 130, 70, 513), (2, 1472, 512, 2048), (1, 16, 13, 7), (2, 128, 1024, 1024),
                                      (1, 130, 200, 516), (2, 24, 128, 260), (1, 8, 96, 4), (2, 40, 129, 252)])
def test_pointwise_gemm_on_the_bf16_matrix_cores(hip, b, ci, co, n):
    """csrc/pointwise_bf16.hip (opt-in, `pw_math = 'bf16x3'`): the exact three-way bf16 split meets the same 1e-5 bar as the
    fp32-MFMA kernels -- forward with bias and BatchNorm epilogue statistics, backward-data -- for ragged K, M and N;
    plain bf16 operands: 4e-3."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, ci, n, generator=g).to(DEV)
    w = (torch.randn(co, ci, generator=g) * 0.1).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    gy = torch.randn(b, co, n, generator=g).to(DEV)
    ref = torch.einsum('oc,bcn->bon', w.double(), x.double()) + bias.double().view(1, -1, 1)
    y, part = hip.pwconv_forward_split(x, w, bias, 3, want_stats=True)
    assert _rel(y, ref) < 1e-5
    centred = (y.double() - bias.double().view(1, -1, 1)).transpose(0, 1).reshape(co, -1)
    sums = part.double().sum(dim=1)
    assert _rel(sums[:, 0], centred.sum(dim=1)) < 1e-5 and _rel(sums[:, 1], (centred * centred).sum(dim=1)) < 1e-5
    assert _rel(hip.pwconv_backward_data_split(gy, w, 3), torch.einsum('oc,bon->bcn', w.double(), gy.double())) < 1e-5
    assert _rel(hip.pwconv_forward_split(x, w, bias, 1), ref) < 4e-3
    # f16x2: scaled fp16 hi + lo, three partial products
    y2, part2 = hip.pwconv_forward_split(x, w, bias, 2, want_stats=True)
    assert _rel(y2, ref) < 1e-5 and torch.equal(hip.pwconv_forward_split(x, w, bias, 2), y2)
    centred = (y2.double() - bias.double().view(1, -1, 1)).transpose(0, 1).reshape(co, -1)
    sums = part2.double().sum(dim=1)
    assert _rel(sums[:, 0], centred.sum(dim=1)) < 1e-5 and _rel(sums[:, 1], (centred * centred).sum(dim=1)) < 1e-5
    assert _rel(hip.pwconv_backward_data_split(gy * 1e-9, w * 1e3, 2), torch.einsum('oc,bon->bcn', w.double() * 1e3, gy.double() * 1e-9)) < 1e-5


@pytest.mark.parametrize('b,ci,co,n', [(2, 9, 64, 4096), (3, 64, 128, 1000), (1, 130, 70, 516), (2, 1472, 512, 1024), (1, 16, 13, 8), (16, 128, 1024, 512),
                                      (2, 512, 256, 1000), (1, 250, 200, 516), (3, 700, 450, 36), (1, 256, 256, 4),
                                      (2, 1472, 512, 16384), (5, 450, 700, 16388)])   # the last two: the 256 x 192 and 256 x 256 tiles
def test_pointwise_backward_weight_f16x2(hip, b, ci, co, n):
    """csrc/pointwise_wgrad_f16.hip: grad_w and grad_bias at the 1e-5 bar (vs fp64), bit-reproducible, ragged M / K / N tails,
    gradients far below fp16's range."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(b, ci, n, generator=g).to(DEV)
    gy = (torch.randn(b, co, n, generator=g) * 1e-8).to(DEV)
    ref = torch.einsum('bon,bcn->oc', gy.double(), x.double())
    gw, gb = hip.pwconv_backward_weight_f16(x, gy, with_bias=True)
    assert _rel(gw, ref) < 1e-5 and _rel(gb, gy.double().sum(dim=(0, 2))) < 1e-5
    assert torch.equal(hip.pwconv_backward_weight_f16(x, gy, hip.absmax_bits(x), hip.absmax_bits(gy)), gw)
    assert _rel(gw, ref) < 4 * _rel(hip.pwconv_backward_weight(x, gy), ref) + 1e-7
