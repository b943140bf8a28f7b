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
Tolerances (stated, per tensor, errors relative to the tensor's largest entry):
  * one PVConv: HIP path vs the fp32 oracle stack <= 1e-4; loss <= 1e-5;
  * whole networks: a third evaluation of the same network in float64 (tests/truth_backend.py: every sum exact, all
    discrete decisions those of the fp32 reference semantics) is the yardstick -- see _check_network.
"""
import contextlib
import os

import pytest
import torch
import torch.nn.functional as tf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_LAYER = 1e-4        # one PVConv: HIP path vs fp32 oracle stack, per tensor
TOL_LOSS = 1e-5
NET_FACTOR = 4.0        # whole networks: HIP path's distance from the fp64 truth <= NET_FACTOR x the fp32 oracle stack's
NET_CAP = 0.3           # and never beyond this (PVCNN++: neighbourhood max-pool winners flip under fp32 noise in every stack)


@contextlib.contextmanager
def cpu_stack(backend):
    """Run pvcnn_amd's Python layers on CPU tensors with `backend` at the native seam; coordinate statistics are the
    reference formulation evaluated in fp32 on the device under test."""
    from pvcnn_amd.modules.functional import backend as seam
    from pvcnn_amd.modules import voxelization as vz
    prev, orig = seam._backend, vz.Voxelization.normalized_coords

    def same_device_coords(self, coords):
        return orig(self, coords.float().to(DEV)).cpu() if not coords.is_cuda else orig(self, coords)
    seam._backend = backend
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


class PinMaxWinners(torch.overrides.TorchFunctionMode):
    """Record (first run) or replay (later runs) the winners of every feature max-pool -- `x.max(dim=...)` on a tensor
    that carries gradient: the global max over the points (models/s3dis/pvcnn.py:43, models/shapenet/pvcnn.py:41) and
    the max over each ball-query neighbourhood (modules/pointnet.py:76).  A winner is a DISCRETE decision between
    values that regularly differ by less than fp32 round-off (2048 candidates per channel: top-2 gaps of 1e-7 relative
    occur in every run, and the 5 % exact duplicate points tie exactly), and the whole gradient of that channel is
    routed to the winner -- so two correct fp32 evaluations of the same network disagree by 1e-3..1e-1 in single
    gradient entries whenever one winner flips (measured here: the fp32 oracle stack vs the fp64 truth, without any
    GPU involved).  Pinning the winners to the fp64 evaluation's removes exactly that and nothing else."""

    def __init__(self, winners=None):
        super().__init__()
        self.replay = winners is not None
        self.winners = winners if self.replay else []
        self.pos = 0

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.Tensor.max and len(args) == 1 and 'dim' in kwargs and args[0].requires_grad:
            x, dim, keep = args[0], kwargs['dim'], kwargs.get('keepdim', False)
            if not self.replay:
                out = func(*args, **kwargs)
                self.winners.append(out.indices.detach().cpu())
                return out
            idx = self.winners[self.pos].to(x.device)
            self.pos += 1
            vals = x.gather(dim, idx if keep else idx.unsqueeze(dim))
            return torch.return_types.max((vals if keep else vals.squeeze(dim), idx))
        return func(*args, **kwargs)


def _outputs(output):
    """The tensors a network RETURNS, by name: `<logits>` for the (B, classes, N) tensor of the segmentation nets
    (models/s3dis/pvcnn.py:46), `<output k>` for every floating-point entry of a tuple / dict (PVConv's pair, Frustum-PVCNN's heads)."""
    if output is None:
        return {}
    if torch.is_tensor(output):
        return {'<logits>': output}
    items = output.items() if isinstance(output, dict) else enumerate(output)
    return {f'<output {k}>': v for k, v in items if torch.is_tensor(v) and v.dtype.is_floating_point}


def is_forward_only(name):
    """Forward-only quantities (what the network returns, BatchNorm running statistics): no allowance for flipped decisions."""
    return name.startswith(('buffer ', '<logits>', '<output '))


def _grads(net, leaf, output=None):
    """name -> (CPU, float64) what the network RETURNED (`output`, round 6: the north_star's "outputs within 1e-5 of reference" is
    held per element at every size the whole-network tests run), the gradient of the input leaf and of every parameter, and the
    float buffers (running stats)."""
    out = dict(_outputs(output))
    out['<input>'] = leaf.grad
    for name, p in net.named_parameters():
        if p.grad is not None:
            out[name] = p.grad
    for name, b in net.named_buffers():
        if b.dtype.is_floating_point:
            out['buffer ' + name] = b
    return {k: v.detach().double().cpu() for k, v in out.items()}


def _scale(ref, name):
    """Magnitude an error in tensor `name` is judged against: its own largest entry, or -- for a bias, whose exact
    gradient is ZERO in front of a train-mode BatchNorm (pure round-off in every stack) -- its layer's weight gradient."""
    m = ref[name].abs().max().item()
    if name.endswith('.bias'):
        sib = name[:-len('bias')] + 'weight'
        if sib in ref:
            m = max(m, ref[sib].abs().max().item())
    return max(m, 1e-30)


def _run_three(build, make_inputs, loss_fn, oracle, pin_winners=False):
    """-> (loss, grads) of the HIP path, of the fp32 oracle stack and of the fp64 truth stack, same weights / inputs.
    pin_winners: the two fp32 runs take the max-pool winners of the fp64 run (PinMaxWinners)."""
    from truth_backend import TruthBackend
    torch.manual_seed(11)
    cpu_net = _no_dropout(build()).train()
    state = {k: v.clone() for k, v in cpu_net.state_dict().items()}
    gpu_net = _no_dropout(build())
    gpu_net.load_state_dict(state)
    gpu_net = gpu_net.to(DEV).train()
    f64_net = _no_dropout(build())
    f64_net.load_state_dict(state)
    f64_net = f64_net.double().train()

    record = PinMaxWinners()
    with cpu_stack(TruthBackend(oracle)), (record if pin_winners else contextlib.nullcontext()):
        inp, leaf, tgt = make_inputs('cpu', torch.float64)
        out_t = f64_net(inp)
        loss_t = loss_fn(out_t, tgt)
        loss_t.backward()
    res_t = (loss_t.item(), _grads(f64_net, leaf, out_t))

    def replay():
        return PinMaxWinners(record.winners) if pin_winners else contextlib.nullcontext()
    with replay():
        inp, leaf, tgt = make_inputs(DEV, torch.float32)
        out_g = gpu_net(inp)
        loss_g = loss_fn(out_g, tgt)
        loss_g.backward()
    torch.cuda.synchronize()
    res_g = (loss_g.item(), _grads(gpu_net, leaf, out_g))
    with cpu_stack(oracle), replay():
        inp, leaf, tgt = make_inputs('cpu', torch.float32)
        out_c = cpu_net(inp)
        loss_c = loss_fn(out_c, tgt)
        loss_c.backward()
    res_c = (loss_c.item(), _grads(cpu_net, leaf, out_c))
    return res_g, res_c, res_t


def _report(label, res_g, res_c, res_t):
    (lg, gg), (lc, gc), (lt, gt) = res_g, res_c, res_t
    assert gg.keys() == gc.keys() == gt.keys()
    rows = []
    for k in gt:
        sc = _scale(gt, k)
        rows.append((k, (gg[k] - gc[k]).abs().max().item() / sc, (gg[k] - gt[k]).abs().max().item() / sc,
                     (gc[k] - gt[k]).abs().max().item() / sc))
    w_gc = max(rows, key=lambda t: t[1]); w_gt = max(rows, key=lambda t: t[2]); w_ct = max(rows, key=lambda t: t[3])
    print(f'[train parity] {label}: loss hip {lg:.7f} oracle-fp32 {lc:.7f} truth-fp64 {lt:.7f}; {len(rows)} tensors; '
          f'worst hip-vs-oracle {w_gc[1]:.2e} ({w_gc[0]}), hip-vs-truth {w_gt[2]:.2e} ({w_gt[0]}), '
          f'oracle-vs-truth {w_ct[3]:.2e} ({w_ct[0]})')
    if os.environ.get('PVCNN_PARITY_VERBOSE'):
        for k, a, b, c in sorted(rows, key=lambda t: -t[2])[:20]:
            print(f'    hip-oracle {a:9.2e}  hip-truth {b:9.2e}  oracle-truth {c:9.2e}  {k}')
    return rows


@pytest.mark.parametrize('cin,cout,r,n,se,normalize', [(9, 32, 16, 2048, False, True), (16, 32, 8, 777, True, True),
                                                      (6, 16, 12, 1024, True, False), (9, 64, 32, 4096, False, True)])
def test_pvconv_train_gradients_match_the_oracle_stack(hip, oracle, cin, cout, r, n, se, normalize):
    """ONE PVConv, train mode: loss, input gradient, every parameter gradient and the updated BatchNorm running
    statistics of the HIP path within 1e-4 (per tensor, relative to the tensor's largest entry) of the oracle stack."""
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd import workload
    b = 3
    x0, _ = workload.make_s3dis_batch(b, n)
    x0 = x0[:, :cin].contiguous() if cin <= 9 else torch.cat([x0, torch.randn(b, cin - 9, n)], dim=1)
    if not normalize:
        x0[:, :3] = x0[:, :3] / 3.0 - 0.4                         # already inside the unit ball
    w = torch.randn(b, cout, n)

    def make(dev, dtype):
        x = x0.clone().to(dev, dtype).requires_grad_()
        return (x, x[:, :3, :]), x, w.to(dev, dtype)

    def loss_fn(out, tgt):
        return (out[0] * tgt).mean() + out[0].square().mean()

    label = f'PVConv({cin}->{cout}, R={r}, N={n}, se={se}, normalize={normalize})'
    res = _run_three(lambda: PVConv(cin, cout, 3, r, with_se=se, normalize=normalize), make, loss_fn, oracle)
    rows = _report(label, *res)
    assert abs(res[0][0] - res[1][0]) <= TOL_LOSS * max(abs(res[1][0]), 1.0)
    bad = [(k, a) for k, a, _, _ in rows if a > TOL_LAYER]
    assert not bad, f'{label}: {bad[:6]}'


FLOOR_T = 1e-4         # per-tensor floor of the strict bar (max-norm, relative to the tensor's largest entry)
FLOOR_FWD = 1e-5       # ... for forward-only quantities (BatchNorm running statistics): no discrete decision is amplified into them
SPARSE_Q = 0.9         # flip allowance: this share of a tensor's ELEMENTS must still meet the strict bar


def _kth(e, q):
    n = e.numel()
    return e.kthvalue(min(n, max(1, int(-(-q * n // 1))))).values.item()


def per_tensor_rows(res_g, res_c, res_t):
    """One dict per tensor: element count, max-norm errors of the HIP path and of the fp32 oracle stack against the fp64 truth (relative
    to the tensor's largest entry, _scale), and the SPARSE_Q quantiles of the same element-wise errors."""
    (_, gg), (_, gc), (_, gt) = res_g, res_c, res_t
    assert gg.keys() == gc.keys() == gt.keys()
    rows = []
    for k in gt:
        sc = _scale(gt, k)
        eh = ((gg[k] - gt[k]).abs() / sc).flatten()
        ec = ((gc[k] - gt[k]).abs() / sc).flatten()
        rows.append({'name': k, 'n': eh.numel(), 'hip': eh.max().item(), 'cpu': ec.max().item(),
                     'hip_q': _kth(eh, SPARSE_Q), 'cpu_q': _kth(ec, SPARSE_Q), 'forward_only': is_forward_only(k)})
    return rows


def judge_per_tensor(label, res, flip_cap, max_allowance_share=0.25):
    """PER TENSOR t (round 5; the round-4 bar was one number per network, set by its worst tensor).  Errors are max-norms relative to
    the tensor's largest entry (_scale); `bar_t` = max(NET_FACTOR x |oracle stack - truth|_t, floor_t), floor_t = 1e-4 (1e-5 for
    forward-only quantities: BatchNorm running statistics).  Every tensor is first held to ITS OWN bar; what is beyond it must fit one
    of two NAMED patterns of a discrete decision (a ReLU, LeakyReLU or max taken the other way on a value within fp32 round-off of
    its threshold -- measured between two CPU evaluations without any GPU, tools/parity_probe.py), and is counted and printed:

        strict       |hip - truth|_t <= bar_t;
        flip site    a GRADIENT tensor whose excess is SPARSE: SPARSE_Q of its elements within bar_t, none beyond `flip_cap`
                     (~ 1/sqrt(elements per channel): what one element moves a random-signed per-channel sum by) -- the layer where a
                     decision flipped: one element of its bias gradient, one row of its weight gradient, one point of `<input>`;
        flip shadow  a gradient tensor with a DENSE excess is admitted only if a flip site exists LATER in forward order (= earlier in
                     backward: the flipped element's gradient passes through every layer in front of it and enters EVERY channel's
                     per-channel sum there as one more term -- all elements of those small tensors move a little), and only within
                     `flip_cap` with SPARSE_Q of its elements within flip_cap / 4.  Seen where layers have few elements per channel:
                     Frustum-PVCNN's box nets (32 x 512 points), PVCNN++'s coarse levels (8 x 16 centres).

    A wrong scale, a missing term, a 10 % error of a whole tensor is dense, has no flip site behind it or exceeds the caps, and FAILS
    whatever the worst tensor of the network does (tests/test_host_logic.py::test_the_per_tensor_judge_...).  Forward-only quantities
    never get an allowance.  Strict must hold for >= 1 - max_allowance_share of the tensors."""
    rows = per_tensor_rows(*res)
    for i, r in enumerate(rows):                       # _grads order: <input>, parameters in registration (= forward) order, buffers
        r['bar'] = max(NET_FACTOR * r['cpu'], FLOOR_FWD if r['forward_only'] else FLOOR_T)
        r['order'] = i
    sites = [r for r in rows if not r['forward_only'] and r['hip'] > r['bar'] and r['hip_q'] <= r['bar'] and r['hip'] <= flip_cap]
    last_site = max((r['order'] for r in sites), default=-1)
    strict, shadow, bad = [], [], []
    for r in rows:
        if r['hip'] <= r['bar']:
            strict.append(r)
        elif r in sites:
            continue
        elif not r['forward_only'] and r['order'] < last_site and r['hip'] <= flip_cap and r['hip_q'] <= flip_cap / 4:
            shadow.append(r)
        else:
            bad.append(r)
    errs = sorted(r['hip'] for r in rows)
    print(f'[per tensor] {label}: {len(rows)} tensors: {len(strict)} within their own bar max({NET_FACTOR:g} x oracle-vs-truth_t, floor_t), '
          f'{len(sites)} flip sites (sparse excess), {len(shadow)} in the shadow of a flip site (dense excess), cap {flip_cap:.1e}; '
          f'{len(bad)} FAILED; hip-vs-truth median {errs[len(errs) // 2]:.2e}, 90th percentile {errs[int(0.9 * len(errs))]:.2e}, worst {errs[-1]:.2e}')
    for r in rows:                                     # what the network RETURNS: one line per tensor, always printed
        if r['forward_only'] and not r['name'].startswith('buffer '):
            print(f"    output {r['name']}: n={r['n']}  hip-vs-truth {r['hip']:.2e}  oracle-vs-truth {r['cpu']:.2e}  bar {r['bar']:.1e}  "
                  f"{'ok' if r['hip'] <= r['bar'] else 'BEYOND'}")
    for cls, members in (('site', sites), ('shadow', shadow), ('FAILED', bad)):
        for r in sorted(members, key=lambda r: -r['hip'])[:12]:
            print(f"    {cls:6s} hip {r['hip']:.2e} (q{int(SPARSE_Q * 100)} {r['hip_q']:.2e})  oracle {r['cpu']:.2e} (q{int(SPARSE_Q * 100)} {r['cpu_q']:.2e})  "
                  f"bar {r['bar']:.1e}  n={r['n']}  {r['name']}")
    dump = os.environ.get('PVCNN_PARITY_DUMP')
    if dump:
        import json
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, ''.join(c if c.isalnum() else '_' for c in label)[:80] + '.json'), 'w') as fh:
            json.dump({'label': label, 'flip_cap': flip_cap, 'rows': rows}, fh)
    assert not bad, f'{label}: beyond the per-tensor bars: ' + ', '.join(f"{r['name']} {r['hip']:.2e} > {r['bar']:.1e}" for r in bad[:6])
    assert len(sites) + len(shadow) <= max_allowance_share * len(rows), \
        f'{label}: {len(sites)} + {len(shadow)} of {len(rows)} tensors needed a flip allowance'
    return rows


def _check_network(label, build, make, loss_fn, oracle, flip_allowance):
    """Whole networks, 10-40 train-mode BatchNorms deep, compared with an fp64 evaluation of the same network
    (tests/truth_backend.py) next to the fp32 oracle stack.

    What limits ANY two fp32 evaluations of these networks (measured without a GPU: torch-CPU with 8 threads vs 1
    thread, tools/parity_probe.py) is not accumulated round-off -- in fp64 the gradients are well conditioned (a 1e-9
    input perturbation moves them by 8e-9) -- but DISCRETE decisions taken on values that sit within round-off of a
    threshold: one ReLU whose pre-activation is ~1e-8 switches on in one evaluation and off in the other, and since a
    per-channel gradient sum runs over only B*N = 4096..8192 random-signed terms, that single element moves the
    channel's bias gradient by ~1/sqrt(B*N) ~ 1e-2 relative, and everything upstream of it by ~1e-3.  (Same for a
    max-pool winner; those are pinned to the fp64 run's here, see PinMaxWinners.  ReLU decisions live inside the fused
    kernels and cannot be pinned.)  Per-layer, where no such flip occurs, the HIP path is within 2e-6 of both
    (test_pvconv_train_gradients_match_the_oracle_stack).  Hence the PER-TENSOR criteria of judge_per_tensor (strict bar of the
    tensor's own; flip sites and their shadows up to flip_allowance ~ 1/sqrt(elements per channel) of the smallest level, counted);
    the loss agrees to 1e-5;
    and as the networks really run (winners not pinned) nothing is beyond NET_CAP."""
    res = _run_three(build, make, loss_fn, oracle, pin_winners=True)
    _report(label + ' [max-pool winners pinned]', *res)
    (lg, _), (lc, _), (lt, _) = res
    assert abs(lg - lt) <= TOL_LOSS * max(abs(lt), 1.0) and abs(lg - lc) <= TOL_LOSS * max(abs(lc), 1.0), (lg, lc, lt)
    judge_per_tensor(label + ' [max-pool winners pinned]', res, flip_allowance)

    res = _run_three(build, make, loss_fn, oracle, pin_winners=False)
    rows = _report(label + ' [as is]', *res)
    (lg, _), (lc, _), (lt, _) = res
    assert abs(lg - lt) <= TOL_LOSS * max(abs(lt), 1.0) and abs(lg - lc) <= TOL_LOSS * max(abs(lc), 1.0), (lg, lc, lt)
    bad = [(k, b) for k, _, b, _ in rows if b > NET_CAP]
    assert not bad, f'{label}: beyond {NET_CAP}: {bad[:6]}'


NETS = {
    'PVCNN': (lambda wl: wl.PVCNN(13, 6, width_multiplier=0.25), lambda wl: wl.make_s3dis_batch(4, 2048)),
    'PVCNN2': (lambda wl: wl.PVCNN2(13, 6, width_multiplier=0.25), lambda wl: wl.make_s3dis_batch(4, 2048)),
    'PVCNNShapeNet': (lambda wl: wl.PVCNNShapeNet(50, 16, 3, width_multiplier=0.25), lambda wl: wl.make_shapenet_batch(4, 1024)),
}


# 1/sqrt(elements per channel at the smallest level): 8192 points (PVCNN), 4096 (ShapeNet), 4 x 16 centres (PVCNN++)
FLIP = {'PVCNN': 1e-2, 'PVCNNShapeNet': 2e-2, 'PVCNN2': 0.25}


@pytest.mark.parametrize('name', list(NETS))
def test_network_train_gradients_match_the_oracle_stack(hip, oracle, name):
    from pvcnn_amd import workload
    build, batch = NETS[name]
    x0, y0 = batch(workload)

    def make(dev, dtype):
        x = x0.clone().to(dev, dtype).requires_grad_()
        return x, x, y0.to(dev)

    _check_network(name, lambda: build(workload), make, tf.cross_entropy, oracle, FLIP[name])


def test_frustum_segmentation_train_gradients_match_the_oracle_stack(hip, oracle):
    """BASELINE configs[4]'s PVConv part (R = 16, 16, 12, 12): the instance-segmentation net of Frustum-PVCNN in fp32."""
    from pvcnn_amd import workload
    in0, y0 = workload.make_frustum_batch(4, 1024)

    def build():
        return workload.FrustumPVCNNE(3, 12, 8, 128, workload.frustum_size_templates(), 1, 0.25).inst_seg_net

    def make(dev, dtype):
        feats = in0['features'].clone().to(dev, dtype).requires_grad_()
        return {'features': feats, 'one_hot_vectors': in0['one_hot_vectors'].to(dev, dtype)}, feats, y0.to(dev)

    _check_network('Frustum-PVCNN segmentation net (R=16,16,12,12)', build, make, tf.cross_entropy, oracle, 2e-2)


@contextlib.contextmanager
def count_native_calls(names):
    """Count calls of the product backend's methods `names` (instance-level wrappers on the live HipBackend object)."""
    from pvcnn_amd.modules.functional import backend as seam
    be, counts = seam._backend, {n: 0 for n in names}
    for n in names:
        orig = getattr(be, n)

        def wrapped(*a, _o=orig, _n=n, **kw):
            counts[_n] += 1
            return _o(*a, **kw)
        setattr(be, n, wrapped)
    try:
        yield counts
    finally:
        for n in names:
            delattr(be, n)


def test_full_width_cfg2_step_runs_the_default_arithmetic_and_matches_the_oracle_stack(hip, oracle):
    """BASELINE configs[1] AS BENCHED: PVCNN 1xC, B = 16, N = 4096 (dropout 0), one train step.  At this size -- and only at this
    size -- the SharedMLP GEMMs cross `pw_split_min_macs` and take the f16x2 kernels (forward, backward-data, backward-weight) through
    autograd, with the absmax hand-over (`x_amax` saved in ctx, the gradient's tagged maximum); the call counter proves those
    routes ran.  Same criteria as the reduced-width networks: loss to 1e-5 against the fp32 oracle stack and the fp64 truth; every
    tensor judged against ITS OWN bar (judge_per_tensor)."""
    from pvcnn_amd import workload
    x0, y0 = workload.make_s3dis_batch(16, 4096)

    def make(dev, dtype):
        x = x0.clone().to(dev, dtype).requires_grad_()
        return x, x, y0.to(dev)

    watched = ['pwconv_gemm_split', 'pwconv_backward_weight_f16', 'pwconv_forward', 'conv3d_igemm_split', 'conv3d_backward_weight_f16',
               'trilinear_devoxelize_bnact_forward', 'avg_voxelize_apply', 'trilinear_devoxelize_backward_apply']
    with count_native_calls(watched) as calls:
        res = _run_three(lambda: workload.PVCNN(13, 6, width_multiplier=1), make, tf.cross_entropy, oracle, pin_winners=True)
    print(f'[train parity] full-width cfg2 native calls: {calls}')
    # forward 128->1024, 1472->512 (+ 512->256 if above the threshold) and their backward-data launches; three f16x2 weight gradients
    assert calls['pwconv_gemm_split'] >= 4 and calls['pwconv_backward_weight_f16'] >= 2, calls
    assert calls['conv3d_igemm_split'] >= 15 and calls['conv3d_backward_weight_f16'] == 8, calls
    assert calls['trilinear_devoxelize_bnact_forward'] == 4 and calls['avg_voxelize_apply'] == 4 and calls['trilinear_devoxelize_backward_apply'] == 4, calls
    _report('PVCNN 1xC B=16 N=4096 (cfg2 as benched) [max-pool winners pinned]', *res)
    (lg, _), (lc, _), (lt, _) = res
    assert abs(lg - lt) <= TOL_LOSS * max(abs(lt), 1.0) and abs(lg - lc) <= TOL_LOSS * max(abs(lc), 1.0), (lg, lc, lt)
    # per tensor: its own bar; sparse excess up to 3e-2 (one flipped ReLU moves a channel sum over B*N = 65536 random-signed terms by
    # 1/256 = 4e-3; in `<input>` it changes ONE POINT's gradient by O(1) of that point's, measured 1e-2 of the tensor's largest entry)
    judge_per_tensor('PVCNN 1xC B=16 N=4096 (cfg2 as benched)', res, flip_cap=3e-2)


# bf16 operands keep 8 bits: 2^-9 = 2e-3 relative per rounded operand of the eight 3x3x3 convolutions (everything else on the path
# stays fp32 under autocast).  Stated bounds; the measured distances are printed (MI355X, round 3: loss 1e-5, median 5.5e-3, 90th
# percentile 0.19, worst 0.28).  Why a WORST tensor can sit at 0.3 while the loss agrees to 1e-5: the gradient of a convolution weight in front of a
# train-mode BatchNorm is a small difference of large terms (BatchNorm cancels the weight's scale direction exactly) -- the fp32
# oracle stack itself is 5e-3 from the fp64 truth on those tensors, i.e. round-off is amplified ~1e5 x there, and 2^-9 operands
# are 3e4 x coarser than fp32's.  Hence median / 90th percentile / sanity bound instead of a uniform per-tensor bar.
BF16_LOSS_TOL = 1e-3
BF16_GRAD_MEDIAN = 2e-2       # median over the tensors, per tensor relative to its largest entry
BF16_GRAD_P90 = 0.25          # 90th percentile (measured 0.19: the ill-conditioned tensors above are a fifth of this small network's)
BF16_GRAD_WORST = 0.6         # sanity


def test_frustum_segmentation_train_step_under_bf16_autocast(hip, oracle):
    """BASELINE configs[4]'s mode: the Frustum-PVCNN segmentation net (every PVConv of that model: R = 16, 16, 12, 12) under
    torch.autocast(bfloat16) -- Conv3d on bf16 MFMA operands with fp32 accumulation -- one train step against the fp32 oracle
    stack and the fp64 truth.  The reference has no reduced-precision path: the tolerance is the stated bf16 one."""
    from pvcnn_amd import workload
    from truth_backend import TruthBackend
    in0, y0 = workload.make_frustum_batch(8, 1024)

    def build():
        return workload.FrustumPVCNNE(3, 12, 8, 128, workload.frustum_size_templates(), 1, 0.5).inst_seg_net

    torch.manual_seed(11)
    cpu_net = _no_dropout(build()).train()
    state = {k: v.clone() for k, v in cpu_net.state_dict().items()}
    gpu_net = _no_dropout(build())
    gpu_net.load_state_dict(state)
    gpu_net = gpu_net.to(DEV).train()
    f64_net = _no_dropout(build())
    f64_net.load_state_dict(state)
    f64_net = f64_net.double().train()

    def make(dev, dtype):
        feats = in0['features'].clone().to(dev, dtype).requires_grad_()
        return {'features': feats, 'one_hot_vectors': in0['one_hot_vectors'].to(dev, dtype)}, feats, y0.to(dev)

    # the global max-pool's winners are pinned to the fp64 run's (PinMaxWinners): at bf16 accuracy (2^-9) the top-2 gap of a
    # channel over 1024 points is crossed in most channels, and a flipped winner re-routes that channel's whole gradient --
    # a property of 8-bit operands, not of the kernels under test (unpinned, the layer in front of the pool differs by 0.4)
    record = PinMaxWinners()
    with cpu_stack(TruthBackend(oracle)), record:
        inp, leaf, tgt = make('cpu', torch.float64)
        loss_t = tf.cross_entropy(f64_net(inp), tgt)
        loss_t.backward()
    res_t = (loss_t.item(), _grads(f64_net, leaf))
    with count_native_calls(['conv3d_igemm_split']) as calls, PinMaxWinners(record.winners):
        inp, leaf, tgt = make(DEV, torch.float32)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss_g = tf.cross_entropy(gpu_net(inp).float(), tgt)
        loss_g.backward()
        torch.cuda.synchronize()
    res_g = (loss_g.item(), _grads(gpu_net, leaf))
    assert calls['conv3d_igemm_split'] >= 15, calls            # the bf16 implicit GEMM ran (forward + backward-data of 8 convolutions)
    with cpu_stack(oracle), PinMaxWinners(record.winners):
        inp, leaf, tgt = make('cpu', torch.float32)
        loss_c = tf.cross_entropy(cpu_net(inp), tgt)
        loss_c.backward()
    res_c = (loss_c.item(), _grads(cpu_net, leaf))
    rows = _report('Frustum-PVCNN segmentation net under autocast(bf16)', res_g, res_c, res_t)
    assert abs(res_g[0] - res_t[0]) <= BF16_LOSS_TOL * max(abs(res_t[0]), 1.0), (res_g[0], res_t[0])
    errs = sorted(b for _, _, b, _ in rows)
    print(f'[train parity] bf16 autocast: loss rel err {abs(res_g[0] - res_t[0]) / max(abs(res_t[0]), 1.0):.2e}; '
          f'per-tensor hip-vs-truth median {errs[len(errs) // 2]:.2e} worst {errs[-1]:.2e}')
    p90 = errs[int(0.9 * len(errs))]
    print(f'[train parity] bf16 autocast: 90th percentile {p90:.2e}')
    assert errs[len(errs) // 2] <= BF16_GRAD_MEDIAN and p90 <= BF16_GRAD_P90 and errs[-1] <= BF16_GRAD_WORST, (errs[len(errs) // 2], p90, errs[-1])
