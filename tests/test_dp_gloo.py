"""Data-parallel harness on CPU: 2 processes, gloo backend, the same GradBucketReducer that runs
over RCCL on the GPUs.  Property: the all-reduced, averaged gradients of the per-rank micro-batches
equal the gradients of one process computing the mean loss over the concatenated batch
(SURVEY.md 8e), for any bucket size; and parameters / buffers start identical on every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from pvcnn_amd.dp import GradBucketReducer, shard_batch


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _net():
    # BatchNorm-free so that shard means compose exactly (BN statistics are per replica by design)
    return nn.Sequential(nn.Conv1d(6, 16, 1), nn.ReLU(), nn.Conv1d(16, 16, 1), nn.ReLU(), nn.Conv1d(16, 5, 1))


def _worker(rank, world, port, bucket_mb, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                 # different initial weights per rank ...
        model = _net()
        reducer = GradBucketReducer(model, bucket_mb=bucket_mb)   # ... made identical by the rank-0 broadcast
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 6, 32, generator=g)
        y = torch.randint(0, 5, (8, 32), generator=g)
        sl = shard_batch(8, world, rank)
        grads = []
        for _ in range(2):                            # two steps: buckets must reset correctly
            reducer.zero_grad()
            loss = nn.functional.cross_entropy(model(x[sl]), y[sl])
            loss.backward()
            reducer.finish()
            grads.append([p.grad.clone() for p in model.parameters()])
        # plain numpy payloads: pickled by value, so nothing refers back to this (exiting) process
        q.put((rank, [p.detach().numpy().copy() for p in model.parameters()],
               [[g.numpy().copy() for g in step] for step in grads], len(reducer.buckets)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('bucket_mb', [8.0, 0.0005])
def test_two_rank_gradients_equal_big_batch(bucket_mb):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_mb, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    as_t = lambda xs: [torch.from_numpy(x) for x in xs]
    (_, params0, grads0, nb0), (_, params1, grads1, nb1) = results
    params0, params1 = as_t(params0), as_t(params1)
    grads0, grads1 = [as_t(s) for s in grads0], [as_t(s) for s in grads1]
    assert nb0 == nb1 and (nb0 > 1 if bucket_mb < 0.01 else nb0 == 1)
    for a, b in zip(params0, params1):
        assert torch.equal(a, b)                      # broadcast from rank 0
    # single-process reference: same weights, whole batch, mean loss
    model = _net()
    with torch.no_grad():
        for p, src in zip(model.parameters(), params0):
            p.copy_(src)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, 32, generator=g)
    y = torch.randint(0, 5, (8, 32), generator=g)
    nn.functional.cross_entropy(model(x), y).backward()
    for step in range(2):
        for p, ga, gb in zip(model.parameters(), grads0[step], grads1[step]):
            assert torch.equal(ga, gb)                # every rank holds the same reduced gradient
            assert torch.allclose(ga, p.grad, atol=1e-6, rtol=1e-5)


def test_single_process_reducer_is_transparent():
    model = _net()
    reducer = GradBucketReducer(model)                # no process group: world = 1, no collective
    x = torch.randn(4, 6, 16)
    reducer.zero_grad()
    model(x).square().mean().backward()
    reducer.finish()
    ref = [p.grad.clone() for p in model.parameters()]
    model2 = _net()
    model2.load_state_dict(model.state_dict())
    model2(x).square().mean().backward()
    for a, p in zip(ref, model2.parameters()):
        assert torch.equal(a, p.grad)
    assert reducer.gradient_bytes == sum(p.numel() for p in model.parameters()) * 4


def test_shard_batch_requires_equal_shards():
    assert shard_batch(16, 4, 1) == slice(4, 8)
    with pytest.raises(ValueError):
        shard_batch(10, 4, 0)


def test_reducer_refuses_a_second_backward_and_supports_no_sync():
    """A second backward() before finish() used to add local gradients on top of an already reduced bucket (silently
    wrong); now it raises, and accumulation goes through no_sync()."""
    import torch.distributed as dist
    from pvcnn_amd.dp import GradBucketReducer
    import tempfile
    store = os.path.join(tempfile.mkdtemp(), 'store')
    if True:
        dist.init_process_group('gloo', init_method=f'file://{store}', rank=0, world_size=1)
        try:
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
            red = GradBucketReducer(net, always_reduce=True)
            x = torch.randn(6, 4)
            red.zero_grad()
            net(x).sum().backward()
            with pytest.raises(RuntimeError, match='already all-reduced'):
                net(x).sum().backward()
            red.finish()
            # accumulation: two micro-batches == one double batch
            red.zero_grad()
            with red.no_sync():
                net(x[:3]).sum().backward()
            net(x[3:]).sum().backward()
            red.finish()
            acc = [p.grad.clone() for p in net.parameters()]
            red.zero_grad()
            net(x).sum().backward()
            red.finish()
            for a, p in zip(acc, net.parameters()):
                assert torch.allclose(a, p.grad, atol=1e-6)
        finally:
            dist.destroy_process_group()


def _pvcnn_worker(rank, world, port, q):
    """The real model on the data-parallel path: PVCNN (0.125xC, BatchNorm and all) on the CPU oracle backend."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle.oracle_backend import OracleBackend
        from pvcnn_amd import workload
        from pvcnn_amd.modules.functional import backend as seam
        seam._backend = OracleBackend()
        torch.manual_seed(50 + rank)
        model = workload.PVCNN(13, 6, width_multiplier=0.125).train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        reducer = GradBucketReducer(model, bucket_mb=0.05)
        x, y = workload.make_s3dis_batch(4, 256)
        sl = shard_batch(4, world, rank)
        reducer.zero_grad()
        nn.functional.cross_entropy(model(x[sl]), y[sl]).backward()
        reducer.finish()
        q.put((rank, {k: v.detach().numpy().copy() for k, v in model.state_dict().items() if 'running' not in k and 'num_batches' not in k},
               [p.grad.numpy().copy() for p in model.parameters()], len(reducer.buckets)))
    finally:
        dist.destroy_process_group()


def test_two_rank_pvcnn_gradients_are_the_mean_of_the_shard_gradients():
    """With BatchNorm the big-batch identity does not hold (statistics are per replica, in the reference's DataParallel as
    well); what must hold: after the all-reduce every rank has the MEAN of the two shards' gradients, each computed with
    that shard's own batch statistics."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_pvcnn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, state0, grads0, nb0), (_, state1, grads1, nb1) = results
    assert nb0 == nb1 and nb0 > 1
    for k in state0:
        assert (state0[k] == state1[k]).all(), k          # broadcast from rank 0
    for ga, gb in zip(grads0, grads1):
        assert (ga == gb).all()                           # every rank holds the same reduced gradient
    from oracle.oracle_backend import OracleBackend
    from pvcnn_amd import workload
    from pvcnn_amd.modules.functional import backend as seam
    prev = seam._backend
    seam._backend = OracleBackend()
    try:
        x, y = workload.make_s3dis_batch(4, 256)
        per_shard = []
        for r in range(world):
            model = workload.PVCNN(13, 6, width_multiplier=0.125).train()
            for m in model.modules():
                if isinstance(m, nn.Dropout):
                    m.p = 0.0
            model.load_state_dict({k: torch.from_numpy(v) for k, v in state0.items()}, strict=False)
            sl = shard_batch(4, world, r)
            nn.functional.cross_entropy(model(x[sl]), y[sl]).backward()
            per_shard.append([p.grad.clone() for p in model.parameters()])
    finally:
        seam._backend = prev
    scale = max(((a + b) / 2).abs().max().item() for a, b in zip(*per_shard))   # biases in front of a BatchNorm have a zero
    #                                                                               true gradient: judged against the global scale
    for g0, a, b in zip(grads0, *per_shard):
        want = (a + b) / 2
        # (the workers run torch-CPU with 2 threads, this process with its default: fp32 summation order differs)
        assert (torch.from_numpy(g0) - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 1e-5 * scale


def test_gradients_are_packed_into_the_flat_buckets_and_unused_parameters_read_zero():
    """zero_grad() drops the gradients (autograd then hands each one over without an accumulate-add); when a bucket is
    complete its gradients are gathered with one multi-tensor copy and `p.grad` become views of the flat buffer -- also
    for parameters that took no part in this backward (zeros) and for buckets completed only by finish()."""
    torch.manual_seed(3)
    used, unused = nn.Linear(5, 7), nn.Linear(3, 2)
    model = nn.ModuleList([used, unused])
    reducer = GradBucketReducer(model, bucket_mb=8.0)
    assert len(reducer.buckets) == 1
    flat = reducer.buckets[0].flat
    x = torch.randn(4, 5)
    for step in range(2):
        reducer.zero_grad()
        assert all(p.grad is None for p in model.parameters())
        used(x).square().sum().backward()
        # the bucket is not complete (two of its four gradients never arrive): finish() packs it
        reducer.finish()
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        for p in model.parameters():
            assert lo <= p.grad.data_ptr() < hi and p.grad.shape == p.shape
        assert unused.weight.grad.abs().max() == 0 and unused.bias.grad.abs().max() == 0
        twin = nn.Linear(5, 7)
        twin.load_state_dict(used.state_dict())
        twin(x).square().sum().backward()
        assert torch.equal(used.weight.grad, twin.weight.grad) and torch.equal(used.bias.grad, twin.bias.grad)
        # bucket order = reverse parameter order; the flat buffer is exactly the concatenation of the views
        want = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))])
        assert torch.equal(flat, want)
    # gradients zeroed IN PLACE (optimizer.zero_grad(set_to_none=False)) keep accumulating into the views: still correct
    for p in model.parameters():
        p.grad.zero_()
    used(x).square().sum().backward()
    reducer.finish()
    assert torch.equal(used.weight.grad, twin.weight.grad)


def _graphed_worker(rank, world, port, q, overlapped):
    """GraphedTrainStep's multi-rank branches on CPU (capture=False: same sequencing as the GPU path).  overlapped: the ordering the
    'graph+collectives' mode captures -- the reducer's hooks launch each full bucket's all-reduce during backward; else the fallback
    -- the hooks only pack, finish() issues every collective in fixed bucket order after forward + backward; then the optimizer."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pvcnn_amd.graph import GraphedTrainStep
        torch.manual_seed(100 + rank)
        model = _net()
        reducer = GradBucketReducer(model, bucket_mb=0.0005)       # several buckets
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 6, 32, generator=g)
        y = torch.randint(0, 5, (8, 32), generator=g)
        sl = shard_batch(8, world, rank)
        step = GraphedTrainStep(model, lambda: nn.functional.cross_entropy(model(x[sl]), y[sl]), opt, reducer, capture=False,
                                capture_collectives=overlapped)
        assert step.collective and reducer.launch_from_hooks is overlapped and step.mode == 'eager' 
        start = [p.detach().numpy().copy() for p in model.parameters()]
        losses = [float(step()) for _ in range(3)]
        # an eager backward after the graphed steps must still be accepted by the reducer (buckets re-armed)
        extra = float(step.eager_step())
        q.put((rank, start, [p.detach().numpy().copy() for p in model.parameters()], losses + [extra], len(reducer.buckets)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('overlapped,world', [(False, 2), (True, 2), (False, 4), (True, 4)])
def test_graphed_step_multi_rank_branch_matches_big_batch_sgd(overlapped, world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_graphed_worker, args=(r, world, port, q, overlapped)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, start0, end0, _, nb0) = results[0]
    for (_, _, end_r, _, nb_r) in results[1:]:
        assert nb0 == nb_r and nb0 > 1
        for a, b in zip(end0, end_r):
            assert (a == b).all()                     # the ranks stay in lock-step
    # single-process reference: 4 SGD steps on the whole batch
    model = _net()
    with torch.no_grad():
        for p, src in zip(model.parameters(), start0):
            p.copy_(torch.from_numpy(src))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, 32, generator=g)
    y = torch.randint(0, 5, (8, 32), generator=g)
    for _ in range(4):
        opt.zero_grad()
        nn.functional.cross_entropy(model(x), y).backward()
        opt.step()
    for p, got in zip(model.parameters(), end0):
        assert torch.allclose(torch.from_numpy(got), p.detach(), atol=1e-5, rtol=1e-4)


def _mixed_capture_worker(rank, world, port, q, failing_rank, demand):
    """GraphedTrainStep._capture_agreed with a capture of the collectives that FAILS ON ONE RANK ONLY (each rank used to
    decide on its own, so the ranks ended up in different modes, posted different collective sequences and hung).  The capture itself
    is stood in for by an object whose replay() issues the same sequencing eagerly -- what is under test is the agreement."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pvcnn_amd.graph import GraphedTrainStep

        class Stub(GraphedTrainStep):
            def _warm_up(self, warmup):
                self.eager_step()

            def _capture(self, whole_step):
                if whole_step and self.collective and rank == failing_rank:
                    raise RuntimeError('stand-in: this rank cannot capture the collectives')
                outer = self

                class Replay:
                    def replay(self):
                        outer.loss = outer._forward_backward()
                        if whole_step:
                            outer.reducer.finish()
                            outer.optimizer.step()
                self.graph, self.whole_step = Replay(), whole_step
                if not whole_step:
                    self.reducer.rearm()

        torch.manual_seed(100 + rank)
        model = _net()
        reducer = GradBucketReducer(model, bucket_mb=0.0005)
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 6, 32, generator=g)
        y = torch.randint(0, 5, (8, 32), generator=g)
        sl = shard_batch(8, world, rank)
        step = Stub.__new__(Stub)
        # (the constructor's CUDA-only prologue is skipped: the fields it sets, then the part under test)
        step.model, step.loss_fn, step.optimizer, step.reducer = model, (lambda: nn.functional.cross_entropy(model(x[sl]), y[sl])), opt, reducer
        import contextlib
        step.autocast, step.collective, step.capture, step.graph, step.loss = contextlib.nullcontext, True, True, None, None
        step.whole_step, step._sync_hyper, step.capture_error, step.bank = False, None, None, None
        step._warm_up(1)
        try:
            step._capture_agreed(True if demand else None)
            outcome = step.mode
        except RuntimeError as exc:
            outcome = 'raised: ' + str(exc)[:60]
            q.put((rank, outcome, None, None))
            return
        losses = [float(step()) for _ in range(3)]
        q.put((rank, outcome, step.capture_error, [p.detach().numpy().copy() for p in model.parameters()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('failing_rank,demand', [(1, False), (0, False), (1, True)])
def test_a_capture_that_fails_on_one_rank_moves_every_rank_to_the_same_mode(failing_rank, demand):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_mixed_capture_worker, args=(r, world, port, q, failing_rank, demand)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])       # (a hang would time out here)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    modes = [r[1] for r in results]
    if demand:                                         # capture_collectives=True: "capture them or raise" -- on EVERY rank
        assert all(m.startswith('raised') for m in modes), modes
        return
    assert modes == ['graph, collectives after replay'] * world, modes
    assert results[failing_rank][2].startswith('RuntimeError') and 'another rank' in results[1 - failing_rank][2]
    for a, b in zip(results[0][3], results[1][3]):
        assert (a == b).all()                          # ... and they trained in lock-step


def test_graphed_step_wants_a_capturable_optimizer_when_it_captures_the_update():
    from pvcnn_amd.graph import GraphedTrainStep
    model = _net()
    reducer = GradBucketReducer(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    with pytest.raises(ValueError, match='capturable'):
        GraphedTrainStep(model, lambda: model(torch.randn(2, 6, 8)).sum(), opt, reducer, capture=True)


def test_flatten_parameters_keeps_values_and_aliases_the_flat_buffers():
    torch.manual_seed(2)
    model = _net()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    reducer = GradBucketReducer(model, bucket_mb=0.0005).flatten_parameters()
    assert len(reducer.buckets) > 1
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])
    for b in reducer.buckets:
        lo, hi = b.pflat.data_ptr(), b.pflat.data_ptr() + b.pflat.numel() * 4
        assert all(lo <= p.data_ptr() < hi for p in b.params)
        b.pflat.add_(1.0)                              # an update of the flat buffer IS an update of the parameters
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k] + 1.0)
    x = torch.randn(2, 6, 8)
    reducer.zero_grad()
    model(x).sum().backward()                          # autograd still works on the re-seated parameters
    reducer.finish()
    assert all(p.grad is not None for p in model.parameters())


def test_bench_dry_collectives_runs_the_n_rank_sequencing_without_a_gpu():
    """`python bench.py --gpus 4 --dry-collectives`: 4 gloo ranks, both orderings of the multi-rank step, lock-step and equal to
    big-batch SGD; one JSON line, exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--dry-collectives'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-400:]
    report = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert report['ok'] and report['ranks'] == 4 and len(report['orderings']) == 2
    assert all(v['ranks_in_lock_step'] and v['gradient_buckets'] > 1 for v in report['orderings'].values())
