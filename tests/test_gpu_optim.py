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


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv1d(7, 33, 1), nn.BatchNorm1d(33), nn.ReLU(), nn.Conv1d(33, 5, 1), nn.Conv3d(3, 4, 3)).to(DEV)


def _fake_grads(nets, gen):
    for ps in zip(*[n.parameters() for n in nets]):
        grad = torch.randn(ps[0].shape, device=DEV, generator=gen)
        for p in ps:
            p.grad = grad.clone()


def test_flat_adam_refuses_a_checkpoint_whose_param_groups_disagree_and_touches_nothing(hip):
    """a torch.optim.Adam checkpoint with several param groups (a no-decay group for biases, say) used to load with the
    FIRST group's hyper-parameters for everything.  Groups that differ are refused BEFORE any moment or counter is written; groups that
    agree load."""
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.optim import FlatAdam
    a, b = _toy(), _toy()
    b.load_state_dict(a.state_dict())
    params = list(a.parameters())
    first, rest = params[:3], params[3:]               # two groups IN PARAMETER ORDER: the checkpoint's state ids line up with the model's
    gen = torch.Generator(device=DEV).manual_seed(2)
    for wd_rest, ok in ((0.0, False), (1e-3, True)):
        ref = torch.optim.Adam([{'params': first, 'weight_decay': 1e-3}, {'params': rest, 'weight_decay': wd_rest}], lr=1e-2)
        _fake_grads([a], gen)
        ref.step()
        red = GradBucketReducer(b, bucket_mb=0.001)
        flat = FlatAdam(red, lr=5e-1)
        before = (flat.step_count.item(), [m.clone() for m in flat.exp_avg])
        if not ok:
            with pytest.raises(ValueError, match='weight_decay'):
                flat.load_state_dict(ref.state_dict())
            assert flat.step_count.item() == before[0] and all(torch.equal(x, y) for x, y in zip(flat.exp_avg, before[1]))
            assert flat.param_groups[0]['lr'] == 5e-1
        else:
            flat.load_state_dict(ref.state_dict())
            assert flat.param_groups[0]['weight_decay'] == 1e-3 and flat.param_groups[0]['lr'] == 1e-2 and flat.step_count.item() == 1
            for p, q in zip(a.parameters(), b.parameters()):
                assert torch.equal(flat.state[q]['exp_avg'], ref.state[p]['exp_avg'])
        red.remove()


def test_flat_adam_is_a_torch_optimizer_with_adams_checkpoint_layout(hip):
    """FlatAdam subclasses torch.optim.Optimizer; its state_dict loads into torch.optim.Adam and vice versa (the
    reference's train.py:186-199 / 249-255 checkpoint the optimizer), and the continued trajectories agree."""
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.optim import FlatAdam
    a, b = _toy(), _toy()
    b.load_state_dict(a.state_dict())
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=1e-3)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for _ in range(3):
        _fake_grads([a], gen)
        ref.step()
    b.load_state_dict(a.state_dict())
    red = GradBucketReducer(b, bucket_mb=0.001)
    flat = FlatAdam(red, lr=5e-1)                                  # wrong hyper-parameters on purpose: the checkpoint's must win
    assert isinstance(flat, torch.optim.Optimizer) and len(red.buckets) > 1
    flat.load_state_dict(ref.state_dict())
    assert flat.param_groups[0]['lr'] == 1e-2 and flat.param_groups[0]['weight_decay'] == 1e-3 and flat.step_count.item() == 3
    for _ in range(3):
        _fake_grads([a, b], gen)
        ref.step()
        red.finish()
        flat.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
    # ... and back: FlatAdam's state_dict into a fresh torch.optim.Adam on a third copy
    c = _toy()
    c.load_state_dict(b.state_dict())
    back = torch.optim.Adam(c.parameters(), lr=1.0)
    back.load_state_dict(flat.state_dict())
    assert back.param_groups[0]['lr'] == 1e-2
    for _ in range(2):
        _fake_grads([b, c], gen)
        red.finish()
        flat.step()
        back.step()
    for p, q in zip(b.parameters(), c.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)


def test_lr_scheduler_reaches_a_captured_flat_adam_update(hip):
    """The learning rate lives in device memory: a torch.optim.lr_scheduler changes param_groups['lr'] on the host, GraphedTrainStep
    copies it to the device block before the next replay, and the CAPTURED update follows (a scalar kernel argument would have been
    frozen at capture time).  lr -> 0 must freeze the parameters; lr back up must move them again."""
    import torch.nn.functional as tf
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    from pvcnn_amd.optim import FlatAdam
    net = _toy()[:4]
    x = torch.randn(4, 7, 64, device=DEV)
    y = torch.randint(0, 5, (4, 64), device=DEV)
    red = GradBucketReducer(net)
    opt = FlatAdam(red, lr=1e-2)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda epoch: (1.0, 0.0, 0.5)[min(epoch, 2)])     # constructing it is the point
    step = GraphedTrainStep(net, lambda: tf.cross_entropy(net(x), y), opt, red, warmup=2)
    assert step.graph is not None
    snap = lambda: [p.detach().clone() for p in net.parameters()]
    p0 = snap(); step(); torch.cuda.synchronize(); p1 = snap()
    assert all(not torch.equal(a, b) for a, b in zip(p0, p1))
    sched.step()                                                   # lr = 0
    assert opt.param_groups[0]['lr'] == 0.0
    step(); torch.cuda.synchronize(); p2 = snap()
    assert all(torch.equal(a, b) for a, b in zip(p1, p2))
    sched.step()                                                   # lr = 5e-3
    step(); torch.cuda.synchronize(); p3 = snap()
    assert all(not torch.equal(a, b) for a, b in zip(p2, p3))


def test_flat_adam_refuses_parameters_that_left_their_bucket(hip):
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.optim import FlatAdam
    net = _toy()
    red = GradBucketReducer(net)
    opt = FlatAdam(red)
    net.double().float()                                           # re-allocates every parameter
    _fake_grads([net], torch.Generator(device=DEV).manual_seed(2))
    red.finish()
    with pytest.raises(RuntimeError, match='flat bucket'):
        opt.step()


def test_a_flat_adam_step_invalidates_armed_weight_images(hip):
    """a pair armed by weight_bank_refresh() but not consumed must not survive an optimizer step that writes the weights
    through raw pointers (no version counter moves)."""
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.optim import FlatAdam
    from pvcnn_amd.modules.functional import backend as seam
    be = seam._backend
    net = nn.Sequential(nn.Conv3d(16, 64, 3, padding=1)).to(DEV)
    red = GradBucketReducer(net)
    opt = FlatAdam(red, lr=1e-1)
    w = net[0].weight
    be.weight_bank_register(net)
    be.conv_weight_images(w, 2)                                    # first sighting: noted as wanted
    be.weight_bank_refresh()                                       # armed, NOT consumed
    _fake_grads([net], torch.Generator(device=DEV).manual_seed(3))
    red.finish()
    opt.step()
    got = be.conv_weight_images(w, 2)                              # must be images of the UPDATED weight
    again = be.conv_weight_images(w, 2)                            # (nothing armed any more: the layer's own launch)
    torch.cuda.synchronize()
    assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])


def test_backward_kernels_write_parameter_gradients_into_the_buckets(hip):
    """functional/_gradslots.py on the GPU path: with a GradBucketReducer the Conv3d / 1x1-convolution / BatchNorm backward kernels
    write their parameter gradients into the flat buckets themselves -- `_Bucket.pack` has nothing to copy for a PVConv -- and every
    gradient is bit-identical to the step whose gradients were gathered by the multi-tensor copy."""
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd.modules.functional import _gradslots
    torch.manual_seed(11)
    net = PVConv(16, 32, 3, 8).to(DEV).train()
    feats = torch.randn(2, 16, 512, device=DEV)
    coords = torch.rand(2, 3, 512, device=DEV)
    red = GradBucketReducer(net)

    def step():
        red.zero_grad()
        out, _ = net((feats, coords))
        out.square().mean().backward()
        red.finish()
        return [p.grad.clone() for p in net.parameters()]

    copied = []
    orig = torch._foreach_copy_
    torch._foreach_copy_ = lambda d, s: (copied.append(len(d)), orig(d, s))[1]
    try:
        in_place = step()
        assert copied == [], copied
        saved, _gradslots._by_ptr = _gradslots._by_ptr, {}           # no slots: the gathering path
        try:
            gathered = step()
        finally:
            _gradslots._by_ptr = saved
        assert sum(copied) == len(in_place)
    finally:
        torch._foreach_copy_ = orig
    for a, b in zip(in_place, gathered):
        assert torch.equal(a, b)
    red.remove()
