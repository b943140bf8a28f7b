"""pvcnn_amd/graph.py: the training step captured in a hipGraph replays to the same losses as the eager step."""
import copy

import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_graphed_step_follows_the_eager_trajectory(hip):
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    torch.manual_seed(0)
    model = workload.PVCNN(13, 6, width_multiplier=0.5).to(DEV).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x, y = workload.make_s3dis_batch(2, 2048, device=DEV, seed=3)
    twin = copy.deepcopy(model)

    def build(mod):
        return GradBucketReducer(mod), torch.optim.Adam(mod.parameters(), lr=1e-3, fused=True, capturable=True)
    red, opt = build(model)

    def eager():
        red.zero_grad()
        loss = tf.cross_entropy(model(x), y)
        loss.backward()
        red.finish()
        opt.step()
        return loss
    want = [eager().item() for _ in range(6)]
    red2, opt2 = build(twin)
    step = GraphedTrainStep(twin, lambda: tf.cross_entropy(twin(x), y), opt2, red2, warmup=3)   # eager steps 0..2 happen in here
    got = [step().item() for _ in range(2)]
    assert want[3] < want[0]                                  # it trains
    # the first replay sees the same weights as the eager twin's step 3 up to round-off; Adam's normalised updates then amplify
    # that round-off (near-zero gradients flip sign), so the bound loosens per step
    for a, b, tol in zip(want[3:], got, (1e-4, 1e-2)):
        assert abs(a - b) <= tol * abs(a), (want, got)
    # new data goes INTO the static input tensors: the replay must see it (the per-coords plans are rebuilt inside the graph)
    x2, y2 = workload.make_s3dis_batch(2, 2048, device=DEV, seed=4)
    ref = copy.deepcopy(twin)
    x.copy_(x2); y.copy_(y2)
    l_graph = step().item()
    l_eager = tf.cross_entropy(ref(x2), y2).item()             # same weights, same batch, eager forward
    assert abs(l_graph - l_eager) <= 1e-4 * abs(l_eager), (l_graph, l_eager)


def test_fused_dropout_draws_a_fresh_mask_on_every_replay(hip):
    """The dropout fused into the BatchNorm passes takes its key from torch.randint: captured in a hipGraph, torch's generator hands
    every replay its own Philox offset, so each replayed step drops different elements (a frozen key would train on one fixed mask)."""
    from pvcnn_amd import workload
    torch.manual_seed(2)
    layers, _ = workload._head(64, [64, 0.5, 13], 1, pointwise=True, classify=True)
    head = torch.nn.Sequential(*layers).to(DEV).train()
    x = torch.randn(2, 64, 1024, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            workload._classify(head, x)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        y = workload._classify(head, x)
    g.replay(); a = y.clone()
    g.replay(); b = y.clone()
    assert torch.isfinite(a).all() and not torch.equal(a, b)


@pytest.mark.parametrize('nsplit', [2, 1])
def test_weight_bank_images_are_the_per_layer_images(hip, nsplit):
    """backend.weight_bank_refresh (one launch per kind for every registered weight; nsplit 2 = the f16x2 pairs, 1 = the plain-bf16
    pairs of the autocast mode) writes byte for byte what the per-layer launches write, arms each pair for exactly one take, and
    never serves a weight changed in place since."""
    import torch.nn as nn
    torch.manual_seed(4)
    net = nn.Sequential(nn.Conv3d(9, 64, 3, padding=1), nn.Conv3d(64, 40, 3, padding=1), nn.Conv1d(70, 130, 1), nn.Conv2d(33, 64, 1),
                        nn.Conv1d(16, 16, 3, padding=1), nn.Linear(4, 4)).to(DEV)
    bank = hip._bank()
    hip.weight_bank_register(net)
    convs = [net[0].weight, net[1].weight]
    pws = [net[2].weight.view(130, 70), net[3].weight.view(64, 33)]
    images = lambda: [hip.conv_weight_images(w, nsplit) for w in convs] + [hip.pw_weight_images(w, nsplit) for w in pws]
    lazy = images()                                                # first sighting: the layers' own launches
    hip.weight_bank_refresh()
    got = images()
    for (af, ab), (bf, bb), w in zip(lazy, got, convs + pws):
        assert torch.equal(af, bf) and torch.equal(ab, bb)
        key = ('conv' if w.dim() == 5 else 'pw', w.data_ptr(), (w.shape[0], w.shape[1]), nsplit)
        assert bank.entries[key]['wf'] is bf                       # served from the bank ...
    again = hip.conv_weight_images(convs[0], nsplit)
    assert again[0] is not got[0][0] and torch.equal(again[0], got[0][0])     # ... once per refresh
    hip.weight_bank_refresh()
    with torch.no_grad():
        net[1].weight.mul_(2.0)                                    # changed in place after the refresh: the armed pair is stale
    fresh = hip.conv_weight_images(convs[1], nsplit)
    assert fresh[0] is not bank.entries[('conv', convs[1].data_ptr(), (40, 64), nsplit)]['wf']
    hip.weight_bank_refresh()
    assert torch.equal(hip.conv_weight_images(convs[1], nsplit)[0], fresh[0])


def test_the_training_loop_of_integration_md_section_c(hip, tmp_path):
    """INTEGRATION.md, section C, end to end on synthetic batches: static input tensors refilled per batch (DevicePrefetcher), the
    captured step (FlatAdam inside), a torch lr_scheduler on top, and a checkpoint that torch.optim.Adam loads."""
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    from pvcnn_amd.optim import FlatAdam
    from pvcnn_amd.pipeline import DevicePrefetcher, synthetic_stream
    torch.manual_seed(1)
    model = workload.PVCNN(13, 6, width_multiplier=0.25).to(DEV).train()
    reducer = GradBucketReducer(model)
    optimizer = FlatAdam(reducer, lr=2e-3, weight_decay=1e-5)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=3)

    def host_batch(i):
        feats, labels = workload.make_s3dis_batch(2, 1024, seed=100 + i)
        return {'features': feats, 'targets': labels}
    first = host_batch(0)
    x, y = first['features'].to(DEV).clone(), first['targets'].to(DEV).clone()
    step = GraphedTrainStep(model, lambda: tf.cross_entropy(model(x), y), optimizer, reducer)
    assert step.mode == 'graph'
    losses, lrs = [], []
    for epoch in range(3):
        for batch in DevicePrefetcher(synthetic_stream(lambda i: host_batch(epoch * 2 + i), 2), DEV):
            x.copy_(batch['features'], non_blocking=True)
            y.copy_(batch['targets'], non_blocking=True)
            losses.append(step())
        scheduler.step()
        lrs.append(optimizer.param_groups[0]['lr'])
    torch.cuda.synchronize()
    vals = [float(losses[-1].detach()), float(losses[0].detach())]
    assert all(torch.isfinite(torch.tensor(vals)))
    assert lrs[0] < 2e-3 and lrs[-1] < lrs[0] and abs(optimizer.hyper[0].item() - lrs[-2]) < 1e-9     # the device block follows the schedule (synced before a replay)
    path = tmp_path / 'ckpt.pt'
    torch.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict()}, path)
    twin = workload.PVCNN(13, 6, width_multiplier=0.25).to(DEV)
    twin.load_state_dict(torch.load(path)['model'])
    adam = torch.optim.Adam(twin.parameters(), lr=1.0)
    adam.load_state_dict(torch.load(path)['optimizer'])
    assert adam.param_groups[0]['lr'] == lrs[-1] and float(adam.state[next(twin.parameters())]['step']) == 9.0      # 3 warm-up steps of the capture + 6 replays
