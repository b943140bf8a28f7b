"""PVCNN++'s furthest-point sampling issued ahead on a stream of its own (pvcnn_amd.workload.centers_ahead) against the in-line order
of the reference (models/s3dis/pvcnnpp.py:44-52: FPS inside each set-abstraction module, modules/pointnet.py:60-62): the same indices
into the same gather, so logits and every gradient are the SAME BITS -- eagerly and as a parallel path of a captured graph."""
import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step(net, x, y):
    x = x.clone().requires_grad_()
    for p in net.parameters():
        p.grad = None
    out = net(x)
    tf.cross_entropy(out, y).backward()
    torch.cuda.synchronize()
    return out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


def _net():
    from pvcnn_amd import workload
    torch.manual_seed(3)
    net = workload.PVCNN2(13, 6, width_multiplier=0.25).to(DEV).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def test_sampling_ahead_is_the_in_line_network_bit_for_bit(hip, monkeypatch):
    from pvcnn_amd import workload
    net = _net()
    x, y = workload.make_s3dis_batch(2, 2048, device=DEV)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    monkeypatch.setattr(workload, '_CENTERS_AHEAD', False)
    out0, gx0, g0 = _step(net, x, y)
    net.load_state_dict(state)                                  # the running statistics moved
    monkeypatch.setattr(workload, '_CENTERS_AHEAD', True)
    out1, gx1, g1 = _step(net, x, y)
    assert torch.equal(out0, out1) and torch.equal(gx0, gx1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # nothing is left behind on the modules, and a set-abstraction module called on its own samples in line
    from pvcnn_amd.modules import PointNetSAModule
    assert not any('_centers_ahead' in m.__dict__ for m in net.modules() if isinstance(m, PointNetSAModule))


def test_a_hand_off_is_taken_only_for_the_tensor_it_was_sampled_from(hip, monkeypatch):
    """The hand-off is valid for exactly the tensor the level before returned, in the state it was returned in (identity + in-place
    version): a module that receives another tensor, or that one modified in place, samples in line -- and so do the levels behind it."""
    from pvcnn_amd import workload
    from pvcnn_amd.modules import PointNetSAModule
    import pvcnn_amd.modules.functional as F
    torch.manual_seed(0)
    sa = PointNetSAModule(num_centers=64, radius=0.3, num_neighbors=8, in_channels=4, out_channels=(8,)).to(DEV).train()
    coords = torch.rand(2, 3, 512, device=DEV)
    feats = torch.randn(2, 4, 512, device=DEV)
    monkeypatch.setattr(workload, '_CENTERS_AHEAD', True)

    # taken: the very tensor
    workload.centers_ahead([sa], coords)
    chain = sa._centers_ahead[2]
    _, centers = sa((feats, coords))
    assert not chain.broken and chain.expected() is centers and torch.equal(centers, F.furthest_point_sample(coords, 64))
    # not taken: an equal copy (another object)
    workload.centers_ahead([sa], coords)
    chain = sa._centers_ahead[2]
    other = coords.clone()
    _, centers = sa((feats, other))
    assert chain.broken and torch.equal(centers, F.furthest_point_sample(other, 64)) and '_centers_ahead' not in sa.__dict__
    # not taken: the tensor itself, modified in place after the sampling was issued
    moved = coords.clone()
    workload.centers_ahead([sa], moved)
    chain = sa._centers_ahead[2]
    moved.mul_(-1.0).add_(torch.rand_like(moved))
    _, centers = sa((feats, moved))
    torch.cuda.synchronize()
    assert chain.broken and torch.equal(centers, F.furthest_point_sample(moved, 64))


def test_sampling_ahead_inside_a_captured_step(hip, monkeypatch):
    """The side stream joins the capture (a parallel path of the graph): the captured step with the sampling ahead trains to the same
    bits as the captured step with the sampling in line."""
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    from pvcnn_amd.optim import FlatAdam
    x, y = workload.make_s3dis_batch(2, 2048, device=DEV)

    def train(ahead):
        monkeypatch.setattr(workload, '_CENTERS_AHEAD', ahead)
        net = _net()
        reducer = GradBucketReducer(net)
        opt = FlatAdam(reducer, lr=1e-3)
        step = GraphedTrainStep(net, lambda: tf.cross_entropy(net(x), y), opt, reducer, warmup=2)
        assert step.mode == 'graph'
        losses = [step().item() for _ in range(3)]
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in net.parameters()]

    losses0, inline = train(False)
    losses1, ahead = train(True)
    assert losses0 == losses1
    for a, b in zip(inline, ahead):
        assert torch.equal(a, b)
