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
