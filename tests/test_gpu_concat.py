"""concat_points (csrc/bnact.hip): torch.cat along the channels + the amax buffer of the result in one kernel; autograd slices."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('b,n', [(3, 1024), (2, 1000), (2, 777), (1, 4)])
def test_concat_points_equals_torch_cat_and_emits_the_amax_buffer(hip, b, n):
    g = torch.Generator().manual_seed(2)
    wide = torch.randn(b, 40, n, generator=g).to(DEV)
    srcs = [torch.randn(b, 9, n, generator=g).to(DEV), wide[:, 4:20, :], torch.randn(b, 130, n, generator=g).to(DEV),
            torch.randn(b, 7, generator=g).to(DEV).unsqueeze(-1).expand(-1, -1, n)]
    srcs[2][0, 5, n // 2] = -321.0
    out, amax = hip.concat_points(srcs)
    want = torch.cat(srcs, dim=1)
    assert torch.equal(out, want)
    assert torch.equal(amax, hip.absmax_tiles(want.contiguous(), 256))
    out2, none = hip.concat_points(srcs, want_amax=False)
    assert none is None and torch.equal(out2, want)


def test_concat_points_autograd_and_tag(hip):
    from pvcnn_amd import workload
    from pvcnn_amd.modules.functional import _cache
    g = torch.Generator().manual_seed(3)
    b, n = 2, 512
    a = torch.randn(b, 8, n, generator=g).to(DEV).requires_grad_()
    c = torch.randn(b, 5, generator=g).to(DEV).requires_grad_()
    a2, c2 = a.detach().clone().requires_grad_(), c.detach().clone().requires_grad_()
    w = torch.randn(b, 13, n, generator=g).to(DEV)
    out = workload.concat_points([a, c.unsqueeze(-1).expand(-1, -1, n)])
    (out * w).sum().backward()
    ref = torch.cat([a2, c2.unsqueeze(-1).expand(-1, -1, n)], dim=1)
    (ref * w).sum().backward()
    assert torch.equal(out, ref) and torch.equal(a.grad, a2.grad) and torch.allclose(c.grad, c2.grad, rtol=1e-5, atol=1e-5)
    tag = _cache.amax_of(out, 256)
    assert tag is not None and torch.equal(tag, hip.absmax_tiles(ref.detach().contiguous(), 256))
