"""csrc/frustum.hip (the box part of Frustum-PointNet's multi-task loss and its gradient, one launch) against the torch formulation
of the same module -- the reference's, modules/frustum.py:43-124 -- in fp64 on the CPU: loss and the gradient of every network
output, relative to each tensor's largest entry."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
KEYS = ('center', 'center_reg', 'heading_scores', 'size_scores', 'heading_residuals_normalized', 'size_residuals_normalized',
        'heading_residuals', 'size_residuals', 'mask_logits')


def _case(b, nh, ns, n, seed, spread):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    templates = torch.rand(ns, 3, generator=g) * 2 + 0.5
    hrn, srn = r(b, nh) * spread, r(b, ns, 3) * spread
    inputs = {'center': r(b, 3) * spread, 'center_reg': r(b, 3) * spread, 'heading_scores': r(b, nh) * 3, 'size_scores': r(b, ns) * 3,
              'heading_residuals_normalized': hrn, 'size_residuals_normalized': srn,
              'heading_residuals': hrn * (math.pi / nh), 'size_residuals': srn * templates.unsqueeze(0), 'mask_logits': r(b, 2, n)}
    targets = {'heading_bin_id': torch.randint(0, nh, (b,), generator=g), 'size_template_id': torch.randint(0, ns, (b,), generator=g),
               'heading_residual': r(b) * 0.2, 'size_residual': r(b, 3) * 0.3, 'center': r(b, 3) * spread,
               'mask_logits': torch.randint(0, 2, (b, n), generator=g)}
    return templates, inputs, targets


@pytest.mark.parametrize('b,nh,ns,n,spread', [(32, 12, 8, 1024, 1.0), (5, 12, 8, 64, 3.0), (300, 4, 3, 16, 0.3), (1, 12, 8, 8, 1.0)])
def test_fused_box_loss_matches_the_torch_formulation(hip, b, nh, ns, n, spread):
    from pvcnn_amd.modules.frustum import FrustumPointNetLoss
    templates, inputs, targets = _case(b, nh, ns, n, 17 * b + nh, spread)
    # fp64 truth on the CPU: the module's torch formulation
    ref = FrustumPointNetLoss(nh, ns, templates.double()).double()
    ref.heading_angle_bin_centers = ref.heading_angle_bin_centers.double()
    ri = {k: v.double().requires_grad_() for k, v in inputs.items()}
    rt = {k: (v.double() if v.is_floating_point() else v) for k, v in targets.items()}
    lr = ref._forward_torch(ri, rt)
    lr.backward()
    # the GPU path
    crit = FrustumPointNetLoss(nh, ns, templates).to(DEV)
    gi = {k: v.to(DEV).requires_grad_() for k, v in inputs.items()}
    gt = {k: v.to(DEV) for k, v in targets.items()}
    from pvcnn_amd.modules.functional._autograd import native
    be, calls = native(), []
    orig = be.frustum_box_loss
    be.frustum_box_loss = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        lg = crit(gi, gt)
    finally:
        del be.frustum_box_loss
    assert calls == [1]                                       # the fused kernel ran
    lg.backward()
    assert abs(lg.item() - lr.item()) <= 2e-6 * max(1.0, abs(lr.item()))
    for k in KEYS:
        a, r = gi[k].grad.double().cpu(), ri[k].grad
        assert (a - r).abs().max().item() <= 1e-5 * max(r.abs().max().item(), 1e-12), k
    # and the torch formulation on the GPU in fp32 agrees with the fused path to rounding
    ti = {k: v.to(DEV).requires_grad_() for k, v in inputs.items()}
    lt = crit._forward_torch(ti, gt)
    lt.backward()
    assert abs(lt.item() - lg.item()) <= 1e-5 * max(1.0, abs(lg.item()))


def test_fused_box_loss_conventions_at_the_kinks(hip):
    """Exact zeros (prediction == target: |v| = 0, |e| = 0) and out-of-the-quadratic-zone errors: the sub-gradients autograd takes."""
    from pvcnn_amd.modules.frustum import FrustumPointNetLoss
    nh, ns, b = 12, 8, 4
    templates, inputs, targets = _case(b, nh, ns, 8, 3, 1.0)
    inputs['center'][0] = targets['center'][0]                # zero-length difference vectors
    inputs['center_reg'][1] = targets['center'][1]
    h, s = targets['heading_bin_id'], targets['size_template_id']
    inputs['heading_residuals_normalized'][2, h[2]] = targets['heading_residual'][2] / (math.pi / nh)      # e == 0 (up to the division's rounding)
    inputs['center'][3] = targets['center'][3] + 50.0         # far outside every Huber zone
    crit = FrustumPointNetLoss(nh, ns, templates).to(DEV)
    gt = {k: v.to(DEV) for k, v in targets.items()}
    gi = {k: v.to(DEV).requires_grad_() for k, v in inputs.items()}
    ti = {k: v.to(DEV).requires_grad_() for k, v in inputs.items()}
    crit(gi, gt).backward()
    crit._forward_torch(ti, gt).backward()
    for k in KEYS:
        assert torch.isfinite(gi[k].grad).all(), k
        assert (gi[k].grad - ti[k].grad).abs().max().item() <= 2e-5 * max(ti[k].grad.abs().max().item(), 1e-12), k
