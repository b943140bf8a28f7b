"""End-to-end: the S3DIS networks on the HIP path vs the same networks on the CPU oracle stack.

PVCNN exercises PVConv (voxelize / Conv3d / devoxelize / SharedMLP); PVCNN++ adds the set-abstraction
and feature-propagation stages (FPS, ball_query, grouping, 3-NN interpolation) with autograd.
Index-producing ops are bit-exact, so both stacks build the same neighbourhoods; features then differ
only by Conv/BatchNorm/GEMM summation order (this package's MFMA kernels vs torch-CPU).  Measured on MI355X (round 3, three
seeds per network, `tools/models_dev.py`): EVERY logit within 1e-7 * (1 + |logit|) of the CPU stack; the bar here is 2e-6, on
every element (rounds 1-2 asserted 99.5 % of the elements within 2e-3)."""
import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _close_fraction(a, b, tol):
    return ((a - b).abs() <= tol * (1 + b.abs())).float().mean().item()


@pytest.mark.parametrize('name,n,batch', [('PVCNN', 2048, 2), ('PVCNN2', 2048, 2)])
def test_network_eval_logits_match_cpu_oracle(hip, oracle, name, n, batch):
    from pvcnn_amd import workload
    from pvcnn_amd.modules.functional import backend as seam
    torch.manual_seed(7)
    cpu_net = getattr(workload, name)(13, 6, width_multiplier=0.25).eval()
    gpu_net = getattr(workload, name)(13, 6, width_multiplier=0.25)
    gpu_net.load_state_dict(cpu_net.state_dict())
    gpu_net = gpu_net.to(DEV).eval()
    x, _ = workload.make_s3dis_batch(batch, n)
    with torch.no_grad():
        got = gpu_net(x.to(DEV)).cpu()
        prev = seam._backend
        seam._backend = oracle
        try:
            want = cpu_net(x)
        finally:
            seam._backend = prev
    assert got.shape == want.shape == (batch, 13, n)
    assert _close_fraction(got, want, 2e-6) == 1.0, ((got - want).abs() / (1 + want.abs())).max().item()


@pytest.mark.parametrize('name,n', [('PVCNN', 2048), ('PVCNN2', 2048)])
def test_network_training_step_is_reproducible(hip, name, n):
    from pvcnn_amd import workload
    torch.manual_seed(3)
    net = getattr(workload, name)(13, 6, width_multiplier=0.25).to(DEV).train()
    for m in net.modules():                      # dropout off: the two passes must then agree exactly
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x, y = workload.make_s3dis_batch(2, n, device=DEV)

    def grads():
        net.zero_grad(set_to_none=True)
        loss = tf.cross_entropy(net(x), y)
        loss.backward()
        return loss.item(), [p.grad.clone() for p in net.parameters() if p.grad is not None]

    bn_state = {k: v.clone() for k, v in net.state_dict().items()}
    l1, g1 = grads()
    net.load_state_dict(bn_state)                # restore BN running stats touched by the first pass
    l2, g2 = grads()
    assert all(torch.isfinite(g).all() for g in g1) and len(g1) > 10
    # every kernel of this package is atomic-free in floating point: the loss and every gradient are BIT-reproducible -- except the
    # weight gradient of the one layer still on the vendor library, the classifier's last plain nn.Conv1d (MIOpen's
    # igemm_wrw_*: global atomics), which moves by ~5e-7 of its largest entry from run to run
    assert l1 == l2
    vendor = {id(m.weight) for m in net.modules() if type(m) is torch.nn.Conv1d and not any(m is c for mlp in net.modules()
              if type(mlp).__name__ == 'SharedMLP' for c in mlp.modules())}
    params = [p for p in net.parameters() if p.grad is not None]
    for p, a, b in zip(params, g1, g2):
        if id(p) in vendor:
            assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()
        else:
            assert torch.equal(a, b), (tuple(p.shape), (a - b).abs().max().item())
