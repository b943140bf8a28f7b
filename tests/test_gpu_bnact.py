"""Fused BatchNorm + ReLU/LeakyReLU (csrc/bnact.hip) against torch's own modules (fp64 reference on
the GPU).  The reference delegates BatchNorm to cuDNN (unpinned summation order): tolerance 1e-5
relative to each tensor's scale, running statistics and num_batches_tracked included."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize('shape,bn_cls,eps,slope', [((4, 64, 16, 16, 16), nn.BatchNorm3d, 1e-4, 0.1),
                                                    ((2, 9, 32, 32, 32), nn.BatchNorm3d, 1e-4, 0.1),
                                                    ((3, 128, 4096), nn.BatchNorm1d, 1e-5, 0.0),
                                                    ((2, 35, 257, 9), nn.BatchNorm2d, 1e-5, 0.0),
                                                    ((2, 7, 1001), nn.BatchNorm1d, 1e-5, 0.0)])
@pytest.mark.parametrize('training', [True, False])
def test_bn_act_matches_torch(hip, shape, bn_cls, eps, slope, training):
    from pvcnn_amd.modules.functional.bnact import run_layers
    torch.manual_seed(0)
    act = nn.LeakyReLU(slope, True) if slope else nn.ReLU(True)
    mine = nn.Sequential(bn_cls(shape[1], eps=eps), act).to(DEV)
    ref = nn.Sequential(bn_cls(shape[1], eps=eps), nn.LeakyReLU(slope) if slope else nn.ReLU()).to(DEV).double()
    with torch.no_grad():
        mine[0].weight.uniform_(0.5, 1.5); mine[0].bias.uniform_(-0.5, 0.5)
        mine[0].running_mean.uniform_(-0.2, 0.2); mine[0].running_var.uniform_(0.5, 1.5)
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in mine.state_dict().items()})
    mine.train(training); ref.train(training)
    x = (torch.randn(*shape, device=DEV) * 2 + 0.7)
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    g = torch.randn(*shape, device=DEV)
    for _ in range(2):      # two steps: running statistics accumulate
        ya = run_layers(mine, xa); yb = ref(xb)
    ya.backward(g); yb.backward(g.double())
    assert _rel(ya, yb.detach()) < 1e-5
    assert _rel(xa.grad, xb.grad) < 2e-5
    assert _rel(mine[0].weight.grad, ref[0].weight.grad) < 2e-5 and _rel(mine[0].bias.grad, ref[0].bias.grad) < 2e-5
    assert _rel(mine[0].running_mean, ref[0].running_mean) < 1e-5 and _rel(mine[0].running_var, ref[0].running_var) < 1e-5
    assert int(mine[0].num_batches_tracked) == int(ref[0].num_batches_tracked)
