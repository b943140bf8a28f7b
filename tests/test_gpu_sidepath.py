"""The weight-gradient launches of backward on their own stream (pvcnn_amd/modules/functional/_sidepath.py) against the in-line order:
the same kernels on the same operands, so every gradient, and a trained step, must be the SAME BITS -- eagerly (two hardware queues)
and as a parallel branch of the captured step.  A race (a block handed out again while the side stream still reads it, a bucket read
before the join) shows up here as a mismatch on some step of the loop."""
import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _net(kind):
    from pvcnn_amd import workload
    torch.manual_seed(11)
    net = (workload.PVCNN(13, 6, width_multiplier=0.5) if kind == 'PVCNN' else workload.PVCNN2(13, 6, width_multiplier=0.25)).to(DEV).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def _train(kind, side, graphed, steps=4):
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    from pvcnn_amd.modules.functional import _sidepath
    from pvcnn_amd.optim import FlatAdam
    keep, _sidepath.enabled = _sidepath.enabled, side
    try:
        net = _net(kind)
        x, y = workload.make_s3dis_batch(2, 2048, device=DEV)
        reducer = GradBucketReducer(net, bucket_mb=0.25)         # several buckets: several joins inside one backward
        opt = FlatAdam(reducer, lr=1e-3)
        forks = []
        orig = _sidepath.forked

        def counting(ref, reads=(), on=True):
            forks.append(bool(on))
            return orig(ref, reads, on)
        _sidepath.forked = counting
        from pvcnn_amd.modules.functional import conv3d, pwconv
        assert conv3d._sidepath is _sidepath and pwconv._sidepath is _sidepath
        try:
            history = []
            if graphed:
                step = GraphedTrainStep(net, lambda: tf.cross_entropy(net(x), y), opt, reducer, warmup=1)
                assert step.mode == 'graph'
                for _ in range(steps):
                    history.append(step().item())
            else:
                for _ in range(steps):
                    reducer.zero_grad()
                    loss = tf.cross_entropy(net(x), y)
                    loss.backward()
                    reducer.finish()
                    history.append((loss.item(), [p.grad.clone() for p in net.parameters()]))
                    opt.step()
        finally:
            _sidepath.forked = orig
        torch.cuda.synchronize()
        assert not _sidepath._open                             # every fork was joined
        return history, [p.detach().clone() for p in net.parameters()], forks
    finally:
        _sidepath.enabled = keep


@pytest.mark.parametrize('kind', ['PVCNN', 'PVCNN2'])
def test_weight_gradients_on_the_side_stream_are_the_in_line_gradients(hip, kind):
    inline, p0, forks0 = _train(kind, False, False)
    side, p1, forks1 = _train(kind, True, False)
    assert forks0 and not any(forks0)                          # switched off: every backward-weight launch in line
    assert sum(forks1) >= 0.9 * len(forks1)                    # switched on: (nearly) all of them take the side stream
    for step, ((l0, g0), (l1, g1)) in enumerate(zip(inline, side)):
        assert l0 == l1, step
        for i, (a, b) in enumerate(zip(g0, g1)):
            assert torch.equal(a, b), (step, i)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def test_the_side_stream_is_a_branch_of_the_captured_step(hip):
    inline, p0, _ = _train('PVCNN', False, True)
    side, p1, forks = _train('PVCNN', True, True)
    assert any(forks)
    assert inline == side
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def test_without_a_gradient_slot_the_launch_stays_in_line(hip):
    """No reducer -> no slot -> autograd would hand the fresh tensor to AccumulateGrad on the main stream: never forked."""
    from pvcnn_amd import workload
    from pvcnn_amd.modules.functional import _sidepath
    net = _net('PVCNN')
    x, y = workload.make_s3dis_batch(2, 1024, device=DEV)
    seen = []
    orig = _sidepath.forked

    def counting(ref, reads=(), on=True):
        seen.append(bool(on))
        return orig(ref, reads, on)
    _sidepath.forked = counting
    try:
        tf.cross_entropy(net(x), y).backward()
    finally:
        _sidepath.forked = orig
    torch.cuda.synchronize()
    assert seen and not any(seen) and not _sidepath._open
