"""The matched bf16 checker of tests/test_gpu_parity_as_benched.py, checked on CPU against plain autograd: with zero recorded
rounding errors and bf16-representable weights it IS the plain convolution (forward, input gradient, weight / bias gradient);
with a recorded perturbation it is the convolution of the perturbed operand whose weight gradient still sees the clean one."""
import torch
import torch.nn as nn

from test_gpu_parity_as_benched import MatchedBf16Convs, _key


def _layer(nd):
    torch.manual_seed(nd)
    m = {1: nn.Conv1d(5, 7, 1), 2: nn.Conv2d(5, 7, 1), 3: nn.Conv3d(5, 7, 3, padding=1)}[nd].double()
    with torch.no_grad():
        m.weight.copy_(m.weight.float().bfloat16().double())
    shape = {1: (2, 5, 9), 2: (2, 5, 3, 4), 3: (2, 5, 4, 4, 4)}[nd]
    return m, torch.randn(*shape, dtype=torch.float64)


def _run(m, x, mode=None):
    x = x.clone().requires_grad_()
    m.zero_grad()
    import contextlib
    with (mode if mode is not None else contextlib.nullcontext()):
        y = m(x)
        (y * torch.linspace(-1, 1, y.numel(), dtype=y.dtype).view_as(y)).sum().backward()
    return y.detach(), x.grad, m.weight.grad.clone(), m.bias.grad.clone()


def test_matched_conv_without_recorded_errors_is_the_plain_convolution():
    for nd in (1, 2, 3):
        m, x = _layer(nd)
        kind = 'conv' if nd == 3 else 'pw'
        want = _run(m, x)
        y = m(x)
        pending = {_key(kind, 'fwd', x.shape, 7): [torch.zeros(x.shape)], _key(kind, 'bwd', y.shape, 5): [torch.zeros(y.shape)]}
        got = _run(m, x, MatchedBf16Convs(pending))
        assert not any(pending.values())                                   # both entries consumed
        for a, b in zip(got, want):
            assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
        # a convolution nothing was recorded for passes through untouched
        got = _run(m, x, MatchedBf16Convs({}))
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_matched_conv_applies_the_recorded_perturbations_to_forward_and_backward_data_only():
    m, x = _layer(3)
    y = m(x)
    dx, dg = torch.randn(x.shape) * 1e-2, torch.randn(y.shape) * 1e-2
    pending = {_key('conv', 'fwd', x.shape, 7): [dx], _key('conv', 'bwd', y.shape, 5): [dg]}
    got_y, got_gx, got_gw, got_gb = _run(m, x, MatchedBf16Convs(pending))
    g = torch.linspace(-1, 1, y.numel(), dtype=y.dtype).view_as(y)
    assert torch.allclose(got_y, m(x + dx.double()).detach(), rtol=1e-12, atol=1e-12)
    xg = x.clone().requires_grad_()
    (m(xg) * (g + dg.double())).sum().backward()
    assert torch.allclose(got_gx, xg.grad, rtol=1e-12, atol=1e-12)          # backward-data: perturbed gradient
    m.zero_grad()
    (m(x) * g).sum().backward()
    assert torch.allclose(got_gw, m.weight.grad, rtol=1e-12, atol=1e-12)    # backward-weight: the clean operands
    assert torch.allclose(got_gb, m.bias.grad, rtol=1e-12, atol=1e-12)
