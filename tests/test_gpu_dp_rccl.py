"""RCCL path on the single GPU of the test box: a 1-rank `nccl` (= RCCL on ROCm) process group with the
reducer forced to issue its collectives.  Sum-all-reduce over one rank is the identity, so gradients
must equal the plain single-process gradients; what this covers is the real RCCL code path (async
all-reduce launched from the autograd hook on flat GPU buckets, stream hand-off, wait) that the
world-size-2 gloo tests cannot."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_bucket_reducer_over_rccl_single_rank(hip):
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(5)
        net = workload.PVCNN(13, 6, width_multiplier=0.25).to(dev).train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        x, y = workload.make_s3dis_batch(2, 1024, device=dev)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        torch.nn.functional.cross_entropy(net(x), y).backward()
        plain = [p.grad.clone() for p in net.parameters()]
        net.zero_grad(set_to_none=True)
        net.load_state_dict(state)
        reducer = GradBucketReducer(net, bucket_mb=0.25, always_reduce=True)   # several buckets -> several collectives
        assert len(reducer.buckets) > 1 and reducer.collective
        reducer.zero_grad()
        torch.nn.functional.cross_entropy(net(x), y).backward()
        reducer.finish()
        torch.cuda.synchronize()
        for p, ref in zip(net.parameters(), plain):
            assert torch.allclose(p.grad, ref, rtol=1e-4, atol=1e-7)
    finally:
        dist.destroy_process_group()


def test_graph_holds_the_rccl_all_reduces_single_rank(hip):
    """pvcnn_amd/graph.py, mode 'graph+collectives': the bucket all-reduces are launched by the reducer's autograd hooks WHILE the
    step is being captured, so the replayed graph contains the RCCL kernels between backward's own (the multi-GPU timed mode of
    bench.py).  One rank (the identity), several buckets: the replayed trajectory follows the eager, collective-free twin's."""
    import copy
    import torch.nn.functional as tf
    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.graph import GraphedTrainStep
    from pvcnn_amd.optim import FlatAdam
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(6)
        model = workload.PVCNN(13, 6, width_multiplier=0.5).to(dev).train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        x, y = workload.make_s3dis_batch(2, 2048, device=dev, seed=3)
        twin = copy.deepcopy(model)
        red = GradBucketReducer(model, bucket_mb=8.0)                        # world 1, no always_reduce: no collective at all
        assert not red.collective
        opt = FlatAdam(red, lr=1e-3)

        def eager():
            red.zero_grad()
            loss = tf.cross_entropy(model(x), y)
            loss.backward()
            red.finish()
            opt.step()
            return loss
        want = [eager().item() for _ in range(6)]
        red2 = GradBucketReducer(twin, bucket_mb=0.5, always_reduce=True)    # several buckets -> several captured collectives
        assert red2.collective and len(red2.buckets) > 2
        opt2 = FlatAdam(red2, lr=1e-3)
        step = GraphedTrainStep(twin, lambda: tf.cross_entropy(twin(x), y), opt2, red2, warmup=3, capture_collectives=True)
        assert step.mode == 'graph+collectives' and step.whole_step and red2.launch_from_hooks
        got = [step().item() for _ in range(3)]
        torch.cuda.synchronize()
        for a, b, tol in zip(want[3:], got, (1e-4, 1e-2, 5e-2)):
            assert abs(a - b) <= tol * abs(a), (want, got)
        # and the fallback ordering (collectives after each replay) still works on the same process group
        twin2 = copy.deepcopy(twin)
        red3 = GradBucketReducer(twin2, bucket_mb=0.5, always_reduce=True)
        step3 = GraphedTrainStep(twin2, lambda: tf.cross_entropy(twin2(x), y), FlatAdam(red3, lr=1e-3), red3, warmup=1,
                                 capture_collectives=False)
        assert step3.mode == 'graph, collectives after replay' and not red3.launch_from_hooks
        assert torch.isfinite(step3()).item()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = torch.device('cuda', rank)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        from pvcnn_amd import workload
        from pvcnn_amd.dp import GradBucketReducer, shard_batch
        torch.manual_seed(9 + rank)
        net = workload.PVCNN(13, 6, width_multiplier=0.25).to(dev).train()
        reducer = GradBucketReducer(net, bucket_mb=0.25)
        x, y = workload.make_s3dis_batch(4, 1024, device=dev)
        sl = shard_batch(4, world, rank)
        reducer.zero_grad()
        torch.nn.functional.cross_entropy(net(x[sl]), y[sl]).backward()
        reducer.finish()
        torch.cuda.synchronize()
        q.put((rank, [p.grad.float().cpu().numpy().copy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs (the 1-GPU test box runs the 1-rank RCCL test above)')
def test_two_rank_rccl_all_reduce_agrees_across_ranks(hip):
    """World size 2 over RCCL / xGMI: after finish() both ranks hold the same gradients (runs on any multi-GPU box)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][1], res[1][1]):
        assert (a == b).all()
