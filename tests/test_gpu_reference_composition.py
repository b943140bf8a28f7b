"""The reference's own model composition (tests/reference_composition.py: plain .max / .repeat / torch.cat / nn.Sequential heads with
nn.Dropout and nn.Conv1d -- models/s3dis/pvcnn.py:34-46, pvcnnpp.py:44-59, shapenet/pvcnn.py:30-42, models/utils.py:15-46) on the GPU,
at the sizes bench.py runs, against the `pvcnn_amd.workload` classes the headline is timed on: same state_dict, same batch, train mode,
dropout p = 0 (the fused dropout draws another stream), forward + backward.

What may differ between the two compositions, and only by fp32 rounding:
  * the classifier's last Conv1d: the vendor library's GEMM (nn.Conv1d as a module) vs this package's kernel (workload._classify) --
    the logits themselves and its three gradients;
  * the backward of `.repeat` (a reshaped sum) vs of `.expand` (sum_to_size) over the N points: two summation orders of 4096 terms;
everything else (the pooled values and winners, the concatenated tensor, the SharedMLP stages) is the same arithmetic on the same bits.
No discrete decision differs: the activations in front of the last Conv1d are bit-identical, so no ReLU and no max-pool winner can
flip -- the bars are fp32 round-off bars, not the whole-network bars of test_gpu_train_parity.py.
"""
import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_FWD = 2e-6          # logits, relative to the largest |logit|
TOL_GRAD = 2e-5         # every gradient tensor, relative to its largest entry (measured values are printed; see the module docstring)
TOL_GRAD_MEDIAN = 2e-6  # ... and the median over the tensors


def _no_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def _scale(ref, name):
    """Magnitude an error in tensor `name` is judged against: its own largest entry, or -- for a bias, whose exact gradient is ZERO in
    front of a train-mode BatchNorm (pure round-off in any evaluation) -- its layer's weight gradient (test_gpu_train_parity._scale)."""
    m = ref[name].abs().max().item()
    if name.endswith('.bias') and name[:-len('bias')] + 'weight' in ref:
        m = max(m, ref[name[:-len('bias')] + 'weight'].abs().max().item())
    return max(m, 1e-30)


def _step(net, x, y):
    x = x.clone().requires_grad_()
    out = net(x)
    loss = tf.cross_entropy(out, y)
    loss.backward()
    torch.cuda.synchronize()
    grads = {'<input>': x.grad.detach().double()}
    grads.update({k: p.grad.detach().double() for k, p in net.named_parameters() if p.grad is not None})
    grads.update({'buffer ' + k: b.detach().double() for k, b in net.named_buffers() if b.dtype.is_floating_point})
    return out.detach().double(), loss.item(), grads


CASES = {
    'cfg2': ('PVCNN', 'ReferencePVCNN', (13, 6), lambda wl: wl.make_s3dis_batch(16, 4096, device=DEV)),
    'cfg3': ('PVCNN2', 'ReferencePVCNN2', (13, 6), lambda wl: wl.make_s3dis_batch(8, 8192, device=DEV)),
    'cfg4': ('PVCNNShapeNet', 'ReferencePVCNNShapeNet', (50, 16, 3), lambda wl: wl.make_shapenet_batch(8, 2048, device=DEV)),
}


@pytest.mark.parametrize('cfg', list(CASES))
def test_reference_composition_equals_the_benched_composition(hip, cfg):
    import reference_composition as rc
    from pvcnn_amd import workload
    mine_name, ref_name, ctor, batch = CASES[cfg]
    torch.manual_seed(7)
    benched = _no_dropout(getattr(workload, mine_name)(*ctor, width_multiplier=1)).to(DEV).train()
    composed = _no_dropout(getattr(rc, ref_name)(*ctor, width_multiplier=1)).to(DEV).train()
    assert list(benched.state_dict().keys()) == list(composed.state_dict().keys())
    composed.load_state_dict(benched.state_dict())
    x, y = batch(workload)
    # (round 5: workload.PVCNN runs its cloud-descriptor head -- Linear + BatchNorm1d + ReLU on 16 rows -- on csrc/dense.hip, another
    #  summation order than the torch modules of the reference composition; that is a difference IN FRONT of the classifier's ReLUs,
    #  i.e. one that can flip them.  It has its own parity test (tests/test_gpu_dense.py); here both sides take the modules, so that what
    #  is compared is the COMPOSITION)
    from pvcnn_amd.modules.functional import dense
    keep = dense._servable
    dense._servable = lambda *a: False
    try:
        out_b, loss_b, g_b = _step(benched, x, y)
    finally:
        dense._servable = keep
    out_c, loss_c, g_c = _step(composed, x, y)
    fwd = ((out_b - out_c).abs().max() / out_b.abs().max()).item()
    assert g_b.keys() == g_c.keys()
    rows = sorted((((g_b[k] - g_c[k]).abs().max().item() / _scale(g_b, k), k) for k in g_b), reverse=True)
    errs = sorted(e for e, _ in rows)
    exact = sum(1 for e in errs if e == 0.0)
    print(f'[reference composition] {cfg}: loss benched {loss_b:.7f} composed {loss_c:.7f}; logits differ by {fwd:.2e} of the largest; '
          f'{len(rows)} tensors: {exact} bit-equal, median {errs[len(errs) // 2]:.2e}, worst {rows[0][0]:.2e} ({rows[0][1]}); '
          f'next: {[(f"{e:.1e}", k) for e, k in rows[1:4]]}')
    import json
    import os
    if os.environ.get('PVCNN_PARITY_DUMP'):
        os.makedirs(os.environ['PVCNN_PARITY_DUMP'], exist_ok=True)
        json.dump({'cfg': cfg, 'fwd': fwd, 'rows': rows}, open(os.path.join(os.environ['PVCNN_PARITY_DUMP'], f'reference_composition_{cfg}.json'), 'w'))
    assert fwd <= TOL_FWD, fwd
    assert abs(loss_b - loss_c) <= 1e-6 * max(abs(loss_b), 1.0)
    # the running statistics are forward-only quantities in front of the last Conv1d: the same bits
    for k in g_b:
        if k.startswith('buffer '):
            assert torch.equal(g_b[k], g_c[k]), k
    bad = [(k, e) for e, k in rows if e > TOL_GRAD]
    assert not bad, bad[:6]
    assert errs[len(errs) // 2] <= TOL_GRAD_MEDIAN


def test_adopt_turns_the_reference_composition_into_the_benched_one(hip):
    """pvcnn_amd.adopt(model): the instance composed as the reference composes it takes workload's forward (and its classifier head the
    module-by-module path) without touching a parameter -- afterwards it IS the benched composition: bit-equal logits and gradients."""
    import pvcnn_amd
    import reference_composition as rc
    from pvcnn_amd import workload
    torch.manual_seed(7)
    benched = _no_dropout(workload.PVCNN(13, 6, width_multiplier=0.5)).to(DEV).train()
    adopted = _no_dropout(rc.ReferencePVCNN(13, 6, width_multiplier=0.5)).to(DEV).train()
    adopted.load_state_dict(benched.state_dict())
    keys = list(adopted.state_dict().keys())
    assert pvcnn_amd.adopt(adopted) is adopted and list(adopted.state_dict().keys()) == keys
    assert type(adopted).forward is workload.PVCNN.forward and type(adopted.classifier)._adopted_from is torch.nn.Sequential
    x, y = workload.make_s3dis_batch(4, 2048, device=DEV)
    out_b, loss_b, g_b = _step(benched, x, y)
    out_a, loss_a, g_a = _step(adopted, x, y)
    assert torch.equal(out_b, out_a) and loss_b == loss_a
    for k in g_b:
        assert torch.equal(g_b[k], g_a[k]), k


ADOPTED = {
    'cfg3 PVCNN2': ('PVCNN2', 'ReferencePVCNN2', (13, 6), lambda wl: wl.make_s3dis_batch(4, 4096, device=DEV)),
    'cfg4 PVCNNShapeNet': ('PVCNNShapeNet', 'ReferencePVCNNShapeNet', (50, 16, 3), lambda wl: wl.make_shapenet_batch(4, 2048, device=DEV)),
}


@pytest.mark.parametrize('case', list(ADOPTED))
def test_adopt_of_the_other_reference_compositions_is_the_benched_composition(hip, case):
    """(round 6) `adopt()` of a PVCNN++ / ShapeNet-PVCNN instance composed the reference's way: workload's forward on the same
    parameters, at FULL width (the launch shapes of the bench line's networks) -- bit-equal logits, loss, gradients, running statistics."""
    import pickle
    import pvcnn_amd
    import reference_composition as rc
    from pvcnn_amd import workload
    mine_name, ref_name, ctor, batch = ADOPTED[case]
    torch.manual_seed(7)
    benched = _no_dropout(getattr(workload, mine_name)(*ctor, width_multiplier=1)).to(DEV).train()
    adopted = _no_dropout(getattr(rc, ref_name)(*ctor, width_multiplier=1)).to(DEV).train()
    adopted.load_state_dict(benched.state_dict())
    keys = list(adopted.state_dict().keys())
    assert pvcnn_amd.adopt(adopted) is adopted and list(adopted.state_dict().keys()) == keys
    assert type(adopted).forward is getattr(workload, mine_name).forward and type(adopted).__name__ == ref_name
    x, y = batch(workload)
    out_b, loss_b, g_b = _step(benched, x, y)
    out_a, loss_a, g_a = _step(adopted, x, y)
    # (the loss is torch's reduction of bit-equal logits: it sums with atomics, the last bit is free)
    assert torch.equal(out_b, out_a) and abs(loss_b - loss_a) <= 1e-6 * abs(loss_b)
    assert g_b.keys() == g_a.keys()
    for k in g_b:
        assert torch.equal(g_b[k], g_a[k]), k
    print(f'[adopt] {case}: logits, loss and {len(g_b)} tensors bit-equal to the benched composition')
    clone = pickle.loads(pickle.dumps(adopted.cpu()))                     # an adopted model pickles
    assert type(clone) is type(adopted) and list(clone.state_dict().keys()) == keys


def _frustum_step(net, inputs, targets, crit, autocast):
    torch.manual_seed(5)                                                  # the device draws of logits_mask: the same stream both times
    feats = inputs['features'].clone().requires_grad_()
    with (torch.autocast('cuda', dtype=torch.bfloat16) if autocast else torch.autocast('cuda', enabled=False)):
        out = net({'features': feats, 'one_hot_vectors': inputs['one_hot_vectors']})
        loss = crit({k: (v.float() if v.dtype.is_floating_point else v) for k, v in out.items()}, targets)
    loss.backward()
    torch.cuda.synchronize()
    grads = {'<input>': feats.grad.detach().double()}
    grads.update({k: p.grad.detach().double() for k, p in net.named_parameters() if p.grad is not None})
    grads.update({'buffer ' + k: b.detach().double() for k, b in net.named_buffers() if b.dtype.is_floating_point})
    return {k: v.detach().double() for k, v in out.items() if v.dtype.is_floating_point}, loss.item(), grads


@pytest.mark.parametrize('autocast', [False, True], ids=['fp32', 'bf16-autocast'])
def test_adopt_of_the_reference_composed_frustum_net_is_the_benched_composition(hip, autocast):
    """(round 6) cfg5: Frustum-PVCNN composed as models/kitti/frustum/ compose it (tests/reference_composition.ReferenceFrustumPVCNNE:
    .repeat / .max / torch.cat, the heads as nn.Sequentials) -> `adopt()` -> bit-equal to workload.FrustumPVCNNE in every returned
    head, the multi-task loss and every gradient, full width, fp32 and under cfg5's autocast; and BEFORE adopt() the reference
    composition itself agrees with it to fp32 round-off in fp32 mode (same operators, other glue)."""
    import pvcnn_amd
    import reference_composition as rc
    from pvcnn_amd import workload
    from pvcnn_amd.modules import FrustumPointNetLoss
    templates = workload.frustum_size_templates()
    ctor = (3, 12, 8, 512, templates, 1, 1)
    torch.manual_seed(7)
    benched = _no_dropout(workload.FrustumPVCNNE(*ctor)).to(DEV).train()
    adopted = _no_dropout(rc.ReferenceFrustumPVCNNE(*ctor)).to(DEV).train()
    assert list(benched.state_dict().keys()) == list(adopted.state_dict().keys())
    adopted.load_state_dict(benched.state_dict())
    inputs, _ = workload.make_frustum_batch(16, 1024, device=DEV)
    targets = workload.make_frustum_targets(16, 1024, device=DEV)
    crit = FrustumPointNetLoss(12, 8, templates).to(DEV)
    out_b, loss_b, g_b = _frustum_step(benched, inputs, targets, crit, autocast)
    if not autocast:
        state = {k: v.clone() for k, v in adopted.state_dict().items()}
        out_r, loss_r, g_r = _frustum_step(adopted, inputs, targets, crit, autocast)     # the reference composition as it is
        adopted.load_state_dict(state)                                                   # (running statistics back to the start)
        adopted.zero_grad(set_to_none=True)                                              # (... and no gradient left to accumulate into)
        dist = {k: ((out_b[k] - out_r[k]).abs().max() / out_b[k].abs().max().clamp_min(1e-30)).item() for k in out_b}
        print(f'[reference composition] cfg5 fp32: loss benched {loss_b:.7f} composed {loss_r:.7f}; heads differ (of the largest) by '
              + ', '.join(f'{k} {v:.1e}' for k, v in dist.items()))
        # the per-point logits are the same operators in another composition: round-off.  (What follows them is held to the same bar only
        # while no point's foreground decision `logits[0] < logits[1]` sits within that round-off -- the sampled points change otherwise.)
        assert dist['mask_logits'] <= 1e-5, dist
    keys = list(adopted.state_dict().keys())
    assert pvcnn_amd.adopt(adopted) is adopted and list(adopted.state_dict().keys()) == keys
    assert type(adopted.inst_seg_net).forward is workload._FrustumSegmentation.forward
    assert type(adopted.center_reg_net).forward is workload._CloudRegressor.forward
    assert type(adopted.box_est_net).forward is workload._CloudRegressor.forward and adopted.box_est_net._coords_tuple
    out_a, loss_a, g_a = _frustum_step(adopted, inputs, targets, crit, autocast)
    assert out_b.keys() == out_a.keys() and g_b.keys() == g_a.keys()
    for k in out_b:
        assert torch.equal(out_b[k], out_a[k]), k
    assert abs(loss_b - loss_a) <= 1e-6 * abs(loss_b)                     # (torch's cross-entropy reduction sums with atomics)
    for k in g_b:
        assert torch.equal(g_b[k], g_a[k]), k
    print(f'[adopt] cfg5 ({"bf16 autocast" if autocast else "fp32"}): {len(out_b)} heads, loss and {len(g_b)} tensors bit-equal to the benched composition')
