"""pvcnn_amd.optim.FlatAdam (csrc/optim.hip) == torch.optim.Adam, step for step, on parameters flattened into the gradient buckets."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('wd,bucket_mb', [(0.0, 8.0), (1e-2, 0.001)])
def test_flat_adam_matches_torch_adam(hip, wd, bucket_mb):
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.optim import FlatAdam
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv1d(7, 33, 1), nn.BatchNorm1d(33), nn.ReLU(), nn.Conv1d(33, 5, 1), nn.Conv3d(3, 4, 3)).to(DEV)
    twin = copy.deepcopy(net)
    red = GradBucketReducer(net, bucket_mb=bucket_mb)
    opt = FlatAdam(red, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ref = torch.optim.Adam(twin.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    assert len(red.buckets) >= (1 if bucket_mb > 1 else 2)
    keys = list(net.state_dict().keys())
    g = torch.Generator(device=DEV).manual_seed(1)
    for step in range(6):
        red.zero_grad()
        ref.zero_grad()
        for p, q in zip(net.parameters(), twin.parameters()):
            grad = torch.randn(p.shape, device=DEV, generator=g)
            p.grad = grad.clone()
            q.grad = grad.clone()
        red.finish()                                   # packs the gradients into the flat buckets
        opt.step()
        ref.step()
        for k, p, q in zip(keys, net.parameters(), twin.parameters()):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), (step, k, (p - q).abs().max().item())
    assert opt.step_count.item() == 6
    assert torch.equal(net.state_dict()['0.weight'], list(net.parameters())[0])      # modules still see their (flattened) parameters


def test_flat_adam_trains_a_graphed_step(hip):
    """The whole step -- zero_grad, forward, loss, backward, FlatAdam -- captured in a hipGraph and replayed: the loss goes down, the
    replays continue the eager trajectory of the SAME optimizer (1e-3 after four steps: the one atomics-based vendor kernel of the step,
    the last Conv1d's backward-weight, is the only difference -- bit-identical on most boxes), and stay near torch.optim.Adam's.  eps = 1e-3: with Adam's default 1e-8 the
    first updates are lr * sign(gradient) for every parameter, gradients at rounding level included, and the trajectories of two
    implementations that round differently drift apart by 1 % within four steps (measured) -- a property of the test net, not of the step."""
    import torch.nn.functional as tf
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    from pvcnn_amd.optim import FlatAdam
    torch.manual_seed(0)
    model = workload.PVCNN(13, 6, width_multiplier=0.25).to(DEV).train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    x, y = workload.make_s3dis_batch(2, 1024, device=DEV, seed=3)
    twin, twin2 = copy.deepcopy(model), copy.deepcopy(model)
    red, red2, red3 = GradBucketReducer(model), GradBucketReducer(twin), GradBucketReducer(twin2)
    opt, opt2, opt3 = FlatAdam(red, lr=1e-3, eps=1e-3), torch.optim.Adam(twin.parameters(), lr=1e-3, eps=1e-3), FlatAdam(red3, lr=1e-3, eps=1e-3)

    def eager(net, reducer, optimizer):
        reducer.zero_grad()
        loss = tf.cross_entropy(net(x), y)
        loss.backward()
        reducer.finish()
        optimizer.step()
        return loss.item()
    want = [eager(twin, red2, opt2) for _ in range(5)]          # torch.optim.Adam
    same = [eager(twin2, red3, opt3) for _ in range(5)]         # FlatAdam, eager
    step = GraphedTrainStep(model, lambda: tf.cross_entropy(model(x), y), opt, red, warmup=3)
    got = [step().item() for _ in range(2)]
    assert want[4] < want[0]
    assert abs(got[0] - same[3]) <= 1e-3 * abs(same[3]) and abs(got[1] - same[4]) <= 1e-3 * abs(same[4]), (same, got)
    assert abs(got[0] - want[3]) <= 2e-3 * abs(want[3]) and abs(got[1] - want[4]) <= 1e-2 * abs(want[4]), (want, got)
