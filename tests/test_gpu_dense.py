"""csrc/dense.hip: nn.Linear + nn.BatchNorm1d + nn.ReLU on a handful of rows (models/utils.py:11-12, the cloud-descriptor heads) as one
launch forward and two backward, against the torch modules it replaces: outputs, every gradient, running statistics and the counter."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 2e-6      # per tensor, relative to its largest entry (two fp32 summation orders of <= 1024 terms)


def _block(cin, cout):
    return nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.ReLU(True))


def _close(a, b, tol=TOL):
    return (a.double() - b.double()).abs().max().item() <= tol * max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize('rows,cin,cout', [(16, 1024, 256), (16, 256, 128), (32, 259, 256), (32, 515, 512), (32, 512, 256), (2, 7, 5), (64, 100, 33), (5, 3, 130)])
def test_linear_bn_relu_matches_the_modules(hip, rows, cin, cout):
    from pvcnn_amd.modules.functional.dense import linear_bn_relu, _servable
    torch.manual_seed(rows * 7 + cin)
    ref = _block(cin, cout).to(DEV).train()
    with torch.no_grad():
        ref[1].weight.uniform_(0.5, 1.5)
        ref[1].bias.normal_()
        ref[1].running_mean.normal_()
        ref[1].running_var.uniform_(0.5, 2.0)
    mine = copy.deepcopy(ref)
    x = torch.randn(rows, cin, device=DEV) * 2 + 0.3
    g = torch.randn(rows, cout, device=DEV)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    assert _servable(mine[0], mine[1], mine[2], xb)
    for step in range(2):                      # two steps: the running statistics and the counter accumulate
        ya = ref(xa)
        yb = linear_bn_relu(mine[0], mine[1], mine[2], xb)
        assert _close(yb, ya), (yb - ya).abs().max().item()
        ya.backward(g)
        yb.backward(g)
    assert mine[1].num_batches_tracked.item() == ref[1].num_batches_tracked.item() == 2
    assert _close(mine[1].running_mean, ref[1].running_mean) and _close(mine[1].running_var, ref[1].running_var)
    # (two rows: x-hat is +-1 and the BatchNorm backward cancels almost everything -- the gradients are ~1e-3 of grad_y, pure round-off of
    #  a difference of O(1) terms: judged against grad_y's scale there)
    assert _close(xb.grad, xa.grad, 1e-5) or (rows == 2 and (xb.grad - xa.grad).abs().max().item() <= 1e-6 * g.abs().max().item() * ref[0].weight.abs().max().item() * cout)
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
        # (the Linear's bias gradient is exactly zero in front of a train-mode BatchNorm: pure round-off, judged against the weight's)
        scale = ref[0].weight.grad.abs().max().item() if n1 == '0.bias' else p1.grad.abs().max().item()
        if rows == 2:
            scale = max(scale, g.abs().max().item() * x.abs().max().item())
        assert (p1.grad - p2.grad).abs().max().item() <= 1e-5 * max(scale, 1e-30), (n1, (p1.grad - p2.grad).abs().max().item(), scale)


def test_run_dense_walks_a_head_and_falls_back(hip):
    """The reference's cloud head [Sequential(Linear, BN1d, ReLU)] x 2 (models/s3dis/pvcnn.py:22-25) and a Frustum regression head
    [block, block, Linear]: fused blocks in training mode (counted), the modules themselves in eval mode; same results either way."""
    from pvcnn_amd.modules.functional.dense import run_dense
    torch.manual_seed(3)
    head = nn.Sequential(_block(259, 256), _block(256, 128), nn.Linear(128, 3)).to(DEV).train()
    twin = copy.deepcopy(head)
    x = torch.randn(32, 259, device=DEV)
    calls = {'n': 0}
    orig = hip.__class__.dense_bn_relu_forward

    def counted(self, *a, **k):
        calls['n'] += 1
        return orig(self, *a, **k)
    hip.__class__.dense_bn_relu_forward = counted
    try:
        got = run_dense(head, x)
        assert calls['n'] == 2
        want = twin(x)
        assert _close(got, want, 1e-5)
        head.eval(); twin.eval()
        # (eval: the modules themselves on both sides; the running statistics the two training passes left differ in the last bit)
        assert _close(run_dense(head, x), twin(x), 1e-5) and calls['n'] == 2
        head.train()
        with torch.autocast('cuda', dtype=torch.bfloat16):        # 16-bit autocast: the modules (torch's autocast rules apply)
            run_dense(head, x)
        assert calls['n'] == 2
    finally:
        hip.__class__.dense_bn_relu_forward = orig


def test_cloud_head_of_pvcnn_runs_fused_and_matches(hip):
    from pvcnn_amd import workload
    torch.manual_seed(5)
    net = workload.PVCNN(13, 6, width_multiplier=0.25).to(DEV).train()
    twin = copy.deepcopy(net)
    for m in list(net.modules()) + list(twin.modules()):
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    x, y = workload.make_s3dis_batch(4, 1024, device=DEV)
    la = nn.functional.cross_entropy(net(x), y)
    # the twin's cloud head through the plain modules
    from pvcnn_amd.modules.functional import dense
    keep = dense._servable
    dense._servable = lambda *a: False
    try:
        lb = nn.functional.cross_entropy(twin(x), y)
        lb.backward()
    finally:
        dense._servable = keep
    la.backward()
    assert abs(la.item() - lb.item()) <= 1e-6 * abs(lb.item())
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), twin.named_parameters()):
        if n1.startswith('cloud_features'):
            sib = dict(twin.named_parameters()).get(n1[:-len('bias')] + 'weight') if n1.endswith('.bias') else None
            scale = max(p2.grad.abs().max().item(), sib.grad.abs().max().item() if sib is not None else 0.0, 1e-30)
            assert (p1.grad - p2.grad).abs().max().item() <= 2e-5 * scale, n1
