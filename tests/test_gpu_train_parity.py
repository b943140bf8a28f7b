"""Train-mode end-to-end parity: loss, input gradient and EVERY parameter gradient of PVConv / PVCNN / PVCNN++ /
ShapeNet-PVCNN / the Frustum segmentation net on the HIP path vs the same network on the CPU oracle stack.

The oracle stack = this package's Python layers with the CPU oracle plugged in at the reference's seam
(`modules.functional.backend._backend`) and torch-CPU Conv/BatchNorm/Linear: it exercises none of the GPU path's
fusion logic (functional/bnact.py's run_layers, BatchNorm statistics from convolution epilogues, BatchNorm+LeakyReLU
folded into the devoxelize gather, strided gradients, the memoised coordinate pre-pass / scatter plans), so agreement
here checks that wiring, not just the kernels one by one.

Two things make the comparison well defined:
  * dropout p = 0 (CPU and GPU RNG streams differ);
  * the voxel coordinates of the checker come from the reference formulation evaluated ON THE DEVICE UNDER TEST
    (`Voxelization.normalized_coords` on cuda:0 -- proven bit-identical to the product path in
    test_gpu_voxel_coords.py): torch's mean / norm reductions differ between CPU and GPU in the last bit, in the
    reference as well, and a point on a rounding boundary would otherwise sit in a different voxel in the two runs.
    Every index tensor downstream (voxel ids, FPS, ball query, 3-NN) is then identical in both stacks.
Tolerance (stated, per tensor): max|a - b| <= TOL * max|b| with TOL = 1e-4 for gradients, 1e-5 for the loss.
"""
import contextlib

import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_GRAD, TOL_LOSS = 1e-4, 1e-5


@contextlib.contextmanager
def oracle_stack(oracle):
    from pvcnn_amd.modules.functional import backend as seam
    from pvcnn_amd.modules import voxelization as vz
    prev, orig = seam._backend, vz.Voxelization.normalized_coords

    def same_device_coords(self, coords):
        return orig(self, coords.to(DEV)).cpu() if not coords.is_cuda else orig(self, coords)
    seam._backend = oracle
    vz.Voxelization.normalized_coords = same_device_coords
    try:
        yield
    finally:
        seam._backend = prev
        vz.Voxelization.normalized_coords = orig


def _no_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _compare(build, make_inputs, loss_fn, label):
    """build() -> fresh module; make_inputs(device) -> (inputs, differentiable leaf, target)."""
    torch.manual_seed(11)
    cpu_net = _no_dropout(build()).train()
    gpu_net = _no_dropout(build())
    gpu_net.load_state_dict(cpu_net.state_dict())
    gpu_net = gpu_net.to(DEV).train()

    inp_g, leaf_g, tgt_g = make_inputs(DEV)
    loss_g = loss_fn(gpu_net(inp_g), tgt_g)
    loss_g.backward()
    torch.cuda.synchronize()
    return cpu_net, gpu_net, leaf_g, loss_g


def _finish(cpu_net, gpu_net, leaf_c, leaf_g, loss_c, loss_g, label):
    worst = ('', 0.0)
    assert abs(loss_c.item() - loss_g.item()) <= TOL_LOSS * max(abs(loss_c.item()), 1.0), (label, loss_c.item(), loss_g.item())
    e = _rel(leaf_g.grad.cpu(), leaf_c.grad)
    assert e <= TOL_GRAD, f'{label}: input gradient rel err {e:.2e}'
    n = 0
    for (name, pc), (_, pg) in zip(cpu_net.named_parameters(), gpu_net.named_parameters()):
        assert (pc.grad is None) == (pg.grad is None), name
        if pc.grad is None:
            continue
        e = _rel(pg.grad.cpu(), pc.grad)
        worst = max(worst, (name, e), key=lambda t: t[1])
        assert e <= TOL_GRAD, f'{label}: grad of {name} rel err {e:.2e}'
        n += 1
    for (name, bc), (_, bg) in zip(cpu_net.named_buffers(), gpu_net.named_buffers()):   # BatchNorm running statistics
        if bc.dtype.is_floating_point:
            assert _rel(bg.cpu(), bc) <= TOL_GRAD, f'{label}: buffer {name}'
    print(f'[train parity] {label}: loss {loss_g.item():.6f} (cpu {loss_c.item():.6f}), {n} parameter gradients, '
          f'worst rel err {worst[1]:.2e} ({worst[0]})')


@pytest.mark.parametrize('cin,cout,r,n,se,normalize', [(9, 32, 16, 2048, False, True), (16, 32, 8, 777, True, True),
                                                      (6, 16, 12, 1024, True, False), (9, 64, 32, 4096, False, True)])
def test_pvconv_train_gradients_match_the_oracle_stack(hip, oracle, cin, cout, r, n, se, normalize):
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd import workload
    b = 3
    x0, _ = workload.make_s3dis_batch(b, n)
    x0 = x0[:, :cin].contiguous() if cin <= 9 else torch.cat([x0, torch.randn(b, cin - 9, n)], dim=1)
    if not normalize:
        x0[:, :3] = x0[:, :3] / 3.0 - 0.4                         # already inside the unit ball
    w = torch.randn(b, cout, n)

    def build():
        return PVConv(cin, cout, 3, r, with_se=se, normalize=normalize)

    def make(dev):
        x = x0.clone().to(dev).requires_grad_()
        return (x, x[:, :3, :]), x, w.to(dev)

    def loss_fn(out, tgt):
        return (out[0] * tgt).mean() + out[0].square().mean()

    cpu_net, gpu_net, leaf_g, loss_g = _compare(build, make, loss_fn, 'PVConv')
    with oracle_stack(oracle):
        inp_c, leaf_c, tgt_c = make('cpu')
        loss_c = loss_fn(cpu_net(inp_c), tgt_c)
        loss_c.backward()
    _finish(cpu_net, gpu_net, leaf_c, leaf_g, loss_c, loss_g, f'PVConv({cin}->{cout}, R={r}, N={n}, se={se}, normalize={normalize})')


NETS = {
    'PVCNN': (lambda wl: wl.PVCNN(13, 6, width_multiplier=0.25), lambda wl, dev: wl.make_s3dis_batch(4, 2048, device=dev)),
    'PVCNN2': (lambda wl: wl.PVCNN2(13, 6, width_multiplier=0.25), lambda wl, dev: wl.make_s3dis_batch(4, 2048, device=dev)),
    'PVCNNShapeNet': (lambda wl: wl.PVCNNShapeNet(50, 16, 3, width_multiplier=0.25),
                      lambda wl, dev: wl.make_shapenet_batch(4, 1024, device=dev)),
}


@pytest.mark.parametrize('name', list(NETS))
def test_network_train_gradients_match_the_oracle_stack(hip, oracle, name):
    from pvcnn_amd import workload
    build, batch = NETS[name]

    def make(dev):
        x, y = batch(workload, dev)
        x = x.clone().requires_grad_()
        return x, x, y

    cpu_net, gpu_net, leaf_g, loss_g = _compare(lambda: build(workload), make, tf.cross_entropy, name)
    with oracle_stack(oracle):
        inp_c, leaf_c, tgt_c = make('cpu')
        loss_c = tf.cross_entropy(cpu_net(inp_c), tgt_c)
        loss_c.backward()
    _finish(cpu_net, gpu_net, leaf_c, leaf_g, loss_c, loss_g, name)


def test_frustum_segmentation_train_gradients_match_the_oracle_stack(hip, oracle):
    """BASELINE configs[4]'s PVConv part (R = 16, 16, 12, 12): the instance-segmentation net of Frustum-PVCNN in fp32."""
    from pvcnn_amd import workload

    def build():
        return workload.FrustumPVCNNE(3, 12, 8, 128, workload.frustum_size_templates(), 1, 0.25).inst_seg_net

    def make(dev):
        inputs, y = workload.make_frustum_batch(4, 1024, device=dev)
        inputs['features'] = inputs['features'].clone().requires_grad_()
        return inputs, inputs['features'], y

    cpu_net, gpu_net, leaf_g, loss_g = _compare(build, make, tf.cross_entropy, 'Frustum seg')
    with oracle_stack(oracle):
        inp_c, leaf_c, tgt_c = make('cpu')
        loss_c = tf.cross_entropy(cpu_net(inp_c), tgt_c)
        loss_c.backward()
    _finish(cpu_net, gpu_net, leaf_c, leaf_g, loss_c, loss_g, 'Frustum-PVCNN segmentation net (R=16,16,12,12)')
