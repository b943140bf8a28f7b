"""Benchmark workloads: the reference networks assembled from pvcnn_amd.modules + synthetic inputs.

The reference's `models/` are callers of the hot path and are NOT part of this package: with
`pvcnn_amd.install_dropin()` they run unchanged on top of `pvcnn_amd.modules`.  bench.py and
smoke() however run on a GPU box where the reference tree does not exist, so the two S3DIS
networks BASELINE.json names (plus the ShapeNet part-segmentation PVCNN of configs[3] and the
Frustum-PVCNN of configs[4]) are assembled here from a small declarative spec.  Parameter
names match the reference classes (models/s3dis/pvcnn.py:9-46, models/s3dis/pvcnnpp.py:8-59,
models/shapenet/pvcnn.py:9-42, models/kitti/frustum/frustum_net.py:14-113, builders
models/utils.py:15-140), so `state_dict`s are interchangeable;
tests/test_reference_python.py checks keys, shapes and outputs against the reference itself.

Synthetic inputs follow SURVEY.md 8(d): seed 1588147245 (configs/__init__.py:3), channel
layout of datasets/s3dis.py:90 (block-local xyz in metres, rgb, room-normalised xyz), ~5 % exact
duplicate points (the loader samples with replacement when a window holds < N points).
"""
import contextlib

import torch
import torch.nn as nn

from .modules import PVConv, PointNetAModule, PointNetFPModule, PointNetSAModule, SharedMLP
from .modules.functional.bnact import emit_row_max, run_layers
from .modules.functional.dense import run_dense

SEED = 1588147245

__all__ = ['PVCNN', 'PVCNN2', 'PVCNNShapeNet', 'FrustumPVCNNE', 'make_s3dis_batch', 'make_shapenet_batch',
           'make_frustum_batch', 'make_frustum_targets', 'frustum_size_templates', 'SEED']


def _scaled(width, k):
    return int(k * width)


class _TapAndPool(torch.autograd.Function):
    """(x (B,C,N), winners (B,C)) -> (x, x[..., winners]): a per-point feature map that is BOTH kept as a tap of the classifier's
    concatenation and max-pooled over the points (models/s3dis/pvcnn.py:41-45) as ONE autograd node.  With two separate consumers
    autograd materialises the pool's gradient as a dense (B,C,N) tensor (a 268 MB zero fill + scatter at PVCNN's 1024 channels) and
    then adds the two gradients with a full pass (another 3 x 268 MB); here the B*C pooled gradients are added into the tap's gradient
    at the winner positions -- the very sum the reference's graph forms, one addition per winner, same rounding."""

    @staticmethod
    def forward(ctx, x, winners, values=None):
        ctx.save_for_backward(winners)
        ctx.npoints = x.shape[-1]
        # values: x[..., winners] already known (the pass that wrote x emitted both: functional/bnact.py: emit_row_max)
        return x.view_as(x), (values.view_as(values) if values is not None else x.gather(2, winners.unsqueeze(-1)).squeeze(-1))

    @staticmethod
    def backward(ctx, g_tap, g_pool):
        (winners,) = ctx.saved_tensors
        if g_tap is None and g_pool is None:
            return None, None, None
        if g_tap is None:
            g = torch.zeros(g_pool.shape + (int(ctx.npoints),), dtype=g_pool.dtype, device=g_pool.device)
        elif getattr(g_tap, '_pvcnn_private_slice', False) and not torch.is_grad_enabled():
            # g_tap is this node's own slice of the concatenation's gradient, handed over by _ConcatPoints.backward (every input gets
            # a disjoint view, nobody else holds it): adding in place keeps it a batch-strided view for the consumer (no copy) and
            # touches B*C elements
            g = g_tap
        else:
            # any other producer (torch.cat's own backward, a hook, an expanded / overlapping tensor, double backward): autograd does
            # not allow modifying a grad_output it may hand to someone else as well
            g = g_tap.clone(memory_format=torch.contiguous_format)
        if g_pool is not None:
            g.scatter_add_(2, winners.unsqueeze(-1), g_pool.unsqueeze(-1).to(g.dtype))
        return g, None, None


class _ConcatPoints(torch.autograd.Function):
    """torch.cat(taps, dim=1) of the classifier input (models/s3dis/pvcnn.py:45) as one kernel that ALSO emits the f16x2 scale table of
    its output (csrc/bnact.hip: concat_points_kernel) -- the first classifier GEMM then needs no pass of its own over the 386 MB it
    reads.  Backward hands every source its channel slice of the gradient as a view (what torch.cat's backward does)."""

    @staticmethod
    def forward(ctx, spec, *taps):
        """spec: None, or (out buffer (B, sum C_i, N), {source index: that source's amax buffer}) -- sources that the pass which
        produced them already wrote into their channel slice of the buffer (emit_row_max(bn, out=...)): nothing is copied for them."""
        from .modules.functional._autograd import native
        be = native()
        ctx.splits = [t.shape[1] for t in taps]
        if spec is None:
            out, amax = be.concat_points([t.detach() for t in taps], want_global=False)      # (its consumers -- the classifier's GEMMs -- take the table)
        else:
            out, amax = be.concat_points([t.detach() for t in taps], out=spec[0], in_place=spec[1], want_global=False)
        ctx.mark_non_differentiable(amax)
        ctx.set_materialize_grads(False)
        return out, amax

    @staticmethod
    def backward(ctx, grad, _grad_amax=None):
        if grad is None:
            return (None,) * (1 + len(ctx.splits))
        grads, off = [None], 0
        for c in ctx.splits:
            piece = grad.narrow(1, off, c)              # a broadcast source: expand's own backward sums its slice over the points
            piece._pvcnn_private_slice = True           # (disjoint views of a buffer only this node holds: _TapAndPool may add in place)
            grads.append(piece)
            off += c
        return tuple(grads)


# PVCNN_CONCAT_SLOT=0 (read once per process): the last stage in a tensor of its own, copied by the concatenation like the others (A/B)
_SLOT_ENABLED = __import__('os').environ.get('PVCNN_CONCAT_SLOT', '1') != '0'


class _Slot:
    """Where the LAST point stage's output goes: its channel slice of the buffer the classifier's concatenation will be, so that
    the BatchNorm + ReLU pass writes it there (268 of PVCNN's 386 MB are then never copied).  Made before that stage runs
    (`concat_slot`), handed to `emit_row_max(bn, out=slot.view)` and to `concat_points(taps, slot=...)`; if the pass did not take
    it (another route: CPU, eval, no row maxima) the tap simply is not in place and is copied like the others."""

    def __init__(self, buffer, index, c0, c1):
        self.buffer, self.index, self.view = buffer, index, buffer[:, c0:c1, :]

    def holds(self, tap):
        v = self.view
        return (tap.data_ptr() == v.data_ptr() and tuple(tap.shape) == tuple(v.shape) and tuple(tap.stride()) == tuple(v.stride()))


def concat_slot(taps_so_far, stage_channels, total_channels, like, training=True):
    """-> _Slot for the stage that comes next (its output: (B, stage_channels, N)) inside a fresh (B, total_channels, N) buffer, or None
    where the GPU path with row maxima is not available -- the same conditions under which the BatchNorm pass takes the row-maximum
    route (train mode, f16x2 products: emit_row_max / batch_norm_act), so that the 386 MB buffer is not allocated for nothing
    (eval mode with grad enabled, fp32 math)."""
    from .modules.functional._autograd import native
    if not (_SLOT_ENABLED and training and like.is_cuda and like.dtype == torch.float32 and torch.is_grad_enabled()):
        return None
    be = native()
    if not (getattr(be, 'has_concat_points', False) and getattr(be, 'has_bnact_rowmax', False)) or like.shape[-1] % 256:
        return None
    if getattr(be, 'pw_math', '') != 'f16x2':
        return None
    c0 = sum(int(t.shape[1]) for t in taps_so_far)
    if c0 + stage_channels > total_channels:
        return None
    buf = torch.empty((like.shape[0], int(total_channels), like.shape[-1]), dtype=torch.float32, device=like.device)
    return _Slot(buf, len(taps_so_far), c0, c0 + int(stage_channels))


def concat_points(taps, slot=None, slot_amax=None):
    """torch.cat(taps, dim=1); on the GPU path one kernel that also tags the result with its f16x2 scale table.  slot / slot_amax: the
    _Slot one of the taps was written into and that tap's amax buffer (concat_slot)."""
    from .modules.functional import _cache
    from .modules.functional._autograd import native
    be = native()
    # (fp32 taps only -- also under torch.autocast, where every tensor between this package's modules is fp32)
    ok = getattr(be, 'has_concat_points', False) and len(taps) <= 8 and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 for t in taps)
    if not ok:
        return torch.cat(taps, dim=1)
    ok = all((t.shape[2] > 1 and t.stride(2) == 0 and (t.stride(1) == 1 or t.shape[1] == 1)) or
             ((t.shape[2] == 1 or t.stride(2) == 1) and (t.shape[1] == 1 or t.stride(1) == t.shape[2])) for t in taps)
    if not ok:
        return torch.cat(taps, dim=1)
    spec = None
    if (slot is not None and slot_amax is not None and slot.index < len(taps) and slot.holds(taps[slot.index])
            and sum(int(t.shape[1]) for t in taps) == slot.buffer.shape[1] and sum(int(t.shape[1]) for t in taps[:slot.index]) * 1 ==
            (slot.view.data_ptr() - slot.buffer.data_ptr()) // (4 * slot.buffer.shape[2])):
        spec = (slot.buffer, {slot.index: slot_amax})
    out, amax = _ConcatPoints.apply(spec, *taps)
    if getattr(be, 'pw_math', '') == 'f16x2':
        _cache.tag_amax(out, be.PW_AMAX_SEG, amax)
    return out


def _out_channels(stage):
    """Output channels of a point stage (PVConv | SharedMLP)."""
    if hasattr(stage, 'out_channels'):
        return int(stage.out_channels)
    convs = [m for m in stage.modules() if isinstance(m, (nn.Conv1d, nn.Conv2d))]
    return int(convs[-1].out_channels)


def _in_channels(head):
    """Input channels of a point-wise head [SharedMLP, ...]: its first convolution's."""
    return int(next(m for m in head.modules() if isinstance(m, (nn.Conv1d, nn.Conv2d))).in_channels)


def _amax_tag(t):
    """The f16x2 scale table the pass that wrote `t` left on it (256-point segments), or None."""
    from .modules.functional import _cache
    from .modules.functional._autograd import native
    return _cache.amax_of(t, getattr(native(), 'PW_AMAX_SEG', 0))


def _last_norm(stage):
    """The BatchNorm of a SharedMLP's last (conv, BatchNorm, ReLU) triple, or None."""
    layers = getattr(stage, 'layers', None)
    if isinstance(layers, nn.Sequential) and len(layers) >= 2 and isinstance(layers[-2], nn.modules.batchnorm._BatchNorm):
        return layers[-2]
    return None


def tap_and_pool(x):
    """-> (x as a tap, max over the points (B,C)).  The winners come from `x.max(dim=-1)` itself (ties: torch's rule)."""
    if not (x.requires_grad and torch.is_grad_enabled()):
        return x, x.max(dim=-1).values
    emitted = getattr(x, '_pvcnn_row_max', None)          # (winners, values, version) from the BatchNorm + ReLU pass that wrote x
    if emitted is not None and emitted[2] == x._version:  # (an in-place op on x since then -- a residual add_, an in-place activation --
        return _TapAndPool.apply(x, emitted[0], emitted[1])   # makes them stale: fall through to a pass over x as it is now)
    from .modules.functional._autograd import native
    be = native() if x.is_cuda else None
    if (be is not None and getattr(be, 'has_neighbor_max', False) and x.dtype == torch.float32 and x.is_contiguous()
            and x.data_ptr() % 16 == 0 and x.shape[-1] % 4 == 0 and x.numel() > 0):
        winners = be.row_argmax(x.detach())              # csrc/pool.hip: one read of x at the streaming rate (same winners)
    else:
        winners = x.max(dim=-1).indices
    return _TapAndPool.apply(x, winners)


def _classify(head, x):
    """head(x) for a point-wise classifier [SharedMLP, Dropout, ..., Conv1d(c, num_classes, 1)] (models/utils.py:15-36) with two things
    the GPU path does differently from nn.Sequential.forward, module by module otherwise:
      * the bare 1x1 Conv1d at the end runs on this package's GEMM kernels like the SharedMLP layers in front of it (as nn.Conv1d it
        is a vendor-library GEMM forward + two GEMMs, a transpose and a reduction backward: ~0.12 ms per PVCNN step for 0.2 GMAC);
      * a training-mode Dropout behind a SharedMLP rides on the passes of that SharedMLP's last BatchNorm + ReLU (csrc/bnact.hip:
        as a module of its own it reads and writes the activated tensor once forward and once backward, 0.15 ms per PVCNN step).
    Eval mode, CPU tensors, hooked modules, autocast: the modules as they are."""
    from .modules.functional.bnact import _has_hooks, fused_dropout_ok
    mods = list(head)
    if _has_hooks(head) or any(_has_hooks(m) for m in mods):
        return head(x)
    i = 0
    while i < len(mods):
        m, nxt = mods[i], (mods[i + 1] if i + 1 < len(mods) else None)
        if (isinstance(m, SharedMLP) and isinstance(nxt, nn.Dropout) and nxt.training and 0.0 < nxt.p < 1.0 and torch.is_tensor(x)
                and fused_dropout_ok(x) and not any(_has_hooks(l) for l in m.layers) and not _has_hooks(m.layers)):
            x = run_layers(m.layers, x, tail_dropout=nxt.p)
            i += 2
        else:
            x = run_layers([m], x)
            i += 1
    return x


def _dense_bn_relu(cin, cout):
    return nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.ReLU(True))


def _head(cin, spec, width, pointwise, classify):
    """MLP head from a spec like [512, 0.3, 256, 0.3, num_classes]: floats < 1 are dropout rates.
    pointwise=True -> SharedMLP / Conv1d on (B,C,N); False -> Linear+BN1d+ReLU on (B,C)."""
    spec = list(spec) if isinstance(spec, (list, tuple)) else [spec]
    block = SharedMLP if pointwise else _dense_bn_relu
    layers = []
    for item in spec[:-1]:
        if item < 1:
            layers.append(nn.Dropout(item))
        else:
            layers.append(block(cin, _scaled(item, width)))
            cin = _scaled(item, width)
    last = spec[-1]
    if classify:
        layers.append(nn.Conv1d(cin, last, 1) if pointwise else nn.Linear(cin, last))
        return layers, last
    layers.append(block(cin, _scaled(last, width)))
    return layers, _scaled(last, width)


def _pv_stack(cin, spec, width, vres, **pvconv_kw):
    """(out_channels, num_blocks, voxel_resolution | None) -> list of PVConv / SharedMLP blocks."""
    cout, repeat, res = spec
    cout = _scaled(cout, width)
    blocks = []
    for _ in range(repeat):
        if res is None:
            blocks.append(SharedMLP(cin, cout))
        else:
            blocks.append(PVConv(cin, cout, kernel_size=3, resolution=int(vres * res), **pvconv_kw))
        cin = cout
    return blocks, cout


class PVCNN(nn.Module):
    """PVCNN for S3DIS semantic segmentation: 4 PVConv + 1 SharedMLP point stages, a global
    max-pooled cloud descriptor, and a point-wise classifier over the concatenation."""
    blocks = ((64, 1, 32), (64, 2, 16), (128, 1, 16), (1024, 1, None))

    def __init__(self, num_classes, extra_feature_channels=6, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        self.in_channels = extra_feature_channels + 3
        stages, cin, concat = [], self.in_channels, 0
        for spec in self.blocks:
            blocks, cin = _pv_stack(cin, spec, width_multiplier, voxel_resolution_multiplier,
                                    with_se=False, normalize=True, eps=0)
            stages += blocks
            concat += cin * len(blocks)
        self.point_features = nn.ModuleList(stages)
        layers, c_cloud = _head(cin, [256, 128], width_multiplier, pointwise=False, classify=False)
        self.cloud_features = nn.Sequential(*layers)
        layers, _ = _head(concat + c_cloud, [512, 0.3, 256, 0.3, num_classes], width_multiplier,
                          pointwise=True, classify=True)
        self.classifier = nn.Sequential(*layers)

    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = inputs['features']
        coords = inputs[:, :3, :]
        feats, taps = inputs, []
        last = len(self.point_features) - 1
        slot = slot_amax = None
        for i, stage in enumerate(self.point_features):
            # (the last stage's BatchNorm + ReLU pass also emits the row maxima the global max-pool needs: no read of that tensor --
            #  and writes its output straight into its slice of the classifier's concatenation: no copy of that tensor either)
            if i == last:
                slot = concat_slot(taps, _out_channels(stage), _in_channels(self.classifier), feats, self.training)
            with (emit_row_max(_last_norm(stage), out=slot.view if slot is not None else None) if i == last else contextlib.nullcontext()):
                feats, _ = stage((feats, coords))
            taps.append(feats)
        if slot is not None and slot.holds(feats):
            slot_amax = _amax_tag(feats)
        taps[-1], pooled = tap_and_pool(feats)             # the last stage's features: a tap AND the global max pool
        cloud = run_dense(self.cloud_features, pooled)    # (Linear + BatchNorm1d + ReLU on 16 rows: one launch per block, csrc/dense.hip)
        # (expand, not repeat: torch.cat reads the broadcast view -- the repeated (B,128,N) tensor is never written on its own)
        taps.append(cloud.unsqueeze(-1).expand(-1, -1, coords.size(-1)))
        return _classify(self.classifier, concat_points(taps, slot, slot_amax))


_CENTERS_AHEAD = __import__('os').environ.get('PVCNN_CENTERS_AHEAD', '1') != '0'
# how many levels of the pyramid are sampled ahead (0 = all): a branch of a replayed graph costs ~3 us per main-chain launch while it
# is open (profiles/ab/r05l), so a level pays only if its chain is long compared with the stage it runs next to
_CENTERS_AHEAD_LEVELS = int(__import__('os').environ.get('PVCNN_CENTERS_AHEAD_LEVELS', '0'))
_side_streams = {}


class _SamplingChain:
    """What ties the levels of `centers_ahead` together: level l was sampled from the centres of level l - 1, so its indices are valid
    for exactly ONE tensor -- the one level l - 1 returned -- in the state it was returned in (identity + in-place version, like
    every entry of functional/_cache.py).  A module that receives anything else (a hook that jitters the coordinates, a caller that
    feeds its own) samples in line and breaks the chain for the levels behind it."""
    __slots__ = ('expected', 'version', 'broken')

    def __init__(self, coords):
        self.broken = False
        self.expect(coords)

    def expect(self, tensor):
        import weakref
        self.expected, self.version = weakref.ref(tensor), tensor._version

    def accepts(self, coords):
        ok = (not self.broken) and self.expected() is coords and coords._version == self.version
        if not ok:
            self.broken = True
        return ok


def centers_ahead(sa_layers, coords):
    """The furthest-point sampling of the whole set-abstraction pyramid, issued NOW on a stream of its own: level l + 1 samples
    from the centres of level l, so the chain depends on the input coordinates alone (pvcnnpp.py:44-52 calls it in front of each
    set-abstraction module, behind that stage's PVConvs).  One workgroup per cloud for M - 1 dependent steps (0.87 + 0.13 ms at
    B = 8, N = 8192 -> 1024 -> 256: 8 of 256 CUs busy) runs next to the first stage's convolutions instead of in front of the
    first set-abstraction module; every module waits for ITS level's event only.  Under a graph capture the side stream joins the
    capture: a parallel path of the graph.  Same indices, same gather -- bit-identical to the in-line order
    (profiles/ab/r05i_sampling_ahead.md: +1.3 %, and the trace of what overlaps).  Each module takes its hand-off only for the very
    tensor the level before it returned (_SamplingChain)."""
    sas = [m for stage in sa_layers for m in (stage if isinstance(stage, nn.Sequential) else [stage]) if isinstance(m, PointNetSAModule)]
    if _CENTERS_AHEAD_LEVELS > 0:
        sas = sas[:_CENTERS_AHEAD_LEVELS]
    if not (_CENTERS_AHEAD and coords.is_cuda and sas):
        return
    from .modules.functional._autograd import native
    be = native()
    main = torch.cuda.current_stream()
    side = _side_streams.get(coords.device)
    if side is None:
        side = _side_streams[coords.device] = torch.cuda.Stream(device=coords.device)
    side.wait_stream(main)
    chain = _SamplingChain(coords)
    with torch.cuda.stream(side), torch.no_grad():
        c = coords.detach()
        for sa in sas:
            picked = be.furthest_point_sampling(c, sa.num_centers)
            done = torch.cuda.Event()
            done.record(side)
            sa._centers_ahead = (picked, done, chain)
            if sa is not sas[-1]:
                c = be.gather_features_forward(c, picked)
    coords.record_stream(side)
    return sas


class PVCNN2(nn.Module):
    """PVCNN++ for S3DIS: PointNet++-style set-abstraction / feature-propagation pyramid whose
    stages are PVConv stacks (ball_query / grouping / FPS / 3-NN interpolation path)."""
    sa_blocks = [
        ((32, 2, 32), (1024, 0.1, 32, (32, 64))),
        ((64, 3, 16), (256, 0.2, 32, (64, 128))),
        ((128, 3, 8), (64, 0.4, 32, (128, 256))),
        (None, (16, 0.8, 32, (256, 256, 512))),
    ]
    fp_blocks = [
        ((256, 256), (256, 1, 8)),
        ((256, 256), (256, 1, 8)),
        ((256, 128), (128, 2, 16)),
        ((128, 128, 64), (64, 1, 32)),
    ]

    def __init__(self, num_classes, extra_feature_channels=6, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        self.in_channels = extra_feature_channels + 3
        k, vr = width_multiplier, voxel_resolution_multiplier
        pv_kw = dict(with_se=True, normalize=True, eps=0)

        # --- set abstraction ----------------------------------------------------------------
        sa_layers, skip_channels = [], []
        extra = extra_feature_channels          # channels handed to the SA module besides xyz
        cin = extra_feature_channels + 3        # channels entering the stage's PVConv stack
        for conv_spec, (n_centers, radius, n_nbrs, widths) in self.sa_blocks:
            skip_channels.append(cin)
            stage = []
            if conv_spec is not None:
                blocks, cin = _pv_stack(cin, conv_spec, k, vr, **pv_kw)
                stage += blocks
                extra = cin
            widths = [[_scaled(w, k) for w in ws] if isinstance(ws, (list, tuple)) else _scaled(ws, k) for ws in widths]
            if n_centers is None:
                sa = PointNetAModule(in_channels=extra, out_channels=widths, include_coordinates=True)
            else:
                sa = PointNetSAModule(num_centers=n_centers, radius=radius, num_neighbors=n_nbrs,
                                      in_channels=extra, out_channels=widths, include_coordinates=True)
            stage.append(sa)
            cin = extra = sa.out_channels
            sa_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
        self.sa_layers = nn.ModuleList(sa_layers)

        # --- feature propagation (raw xyz is dropped from the last skip: pvcnnpp.py:38) -------
        skip_channels[0] = extra_feature_channels
        fp_layers = []
        for i, (fp_widths, conv_spec) in enumerate(self.fp_blocks):
            widths = tuple(_scaled(w, k) for w in fp_widths)
            stage = [PointNetFPModule(in_channels=cin + skip_channels[-1 - i], out_channels=widths)]
            cin = widths[-1]
            if conv_spec is not None:
                blocks, cin = _pv_stack(cin, conv_spec, k, vr, **pv_kw)
                stage += blocks
            fp_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
        self.fp_layers = nn.ModuleList(fp_layers)

        layers, _ = _head(cin, [128, 0.5, num_classes], k, pointwise=True, classify=True)
        self.classifier = nn.Sequential(*layers)

    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = inputs['features']
        coords, feats = inputs[:, :3, :].contiguous(), inputs
        coords_pyramid, skips = [], []
        ahead = centers_ahead(self.sa_layers, coords)
        try:
            for stage in self.sa_layers:
                skips.append(feats)
                coords_pyramid.append(coords)
                feats, coords = stage((feats, coords))
        finally:
            for sa in ahead or ():                    # a forward that raised half-way leaves nothing behind for the next one
                sa.__dict__.pop('_centers_ahead', None)
        skips[0] = inputs[:, 3:, :].contiguous()
        for i, stage in enumerate(self.fp_layers):
            feats, coords = stage((coords_pyramid[-1 - i], coords, feats, skips[-1 - i]))
        return _classify(self.classifier, feats)


class PVCNNShapeNet(nn.Module):
    """PVCNN for ShapeNet part segmentation (BASELINE configs[3]; reference models/shapenet/pvcnn.py:9-42):
    three PVConvs WITH squeeze-excitation on coordinates that are already in the unit ball (normalize=False),
    two SharedMLP point stages, and a classifier over [shape one-hot, every stage's output, global max]."""
    blocks = ((64, 1, 32), (128, 2, 16), (512, 1, None), (2048, 1, None))

    def __init__(self, num_classes, num_shapes, extra_feature_channels=3, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        self.in_channels = extra_feature_channels + 3
        self.num_shapes = num_shapes
        stages, cin, concat = [], self.in_channels, 0
        for spec in self.blocks:
            blocks, cin = _pv_stack(cin, spec, width_multiplier, voxel_resolution_multiplier,
                                    with_se=True, normalize=False, eps=0)
            stages += blocks
            concat += cin * len(blocks)
        self.point_features = nn.ModuleList(stages)
        layers, _ = _head(num_shapes + cin + concat, [256, 0.2, 256, 0.2, 128, num_classes], width_multiplier,
                          pointwise=True, classify=True)
        self.classifier = nn.Sequential(*layers)

    def forward(self, inputs):
        # inputs (B, in_channels + num_shapes, N): xyz, extra features, then the shape one-hot repeated over N
        feats = inputs[:, :self.in_channels, :]
        taps = [inputs[:, -self.num_shapes:, :]]
        coords = feats[:, :3, :]
        last = len(self.point_features) - 1
        slot = slot_amax = None
        for i, stage in enumerate(self.point_features):
            # (the last stage's BatchNorm + ReLU pass also emits the row maxima the global max-pool needs and writes its output into its
            #  slice of the classifier's concatenation: see PVCNN.forward)
            if i == last:
                slot = concat_slot(taps, _out_channels(stage), _in_channels(self.classifier), feats, self.training)
            with (emit_row_max(_last_norm(stage), out=slot.view if slot is not None else None) if i == last else contextlib.nullcontext()):
                feats, _ = stage((feats, coords))
            taps.append(feats)
        if slot is not None and slot.holds(feats):
            slot_amax = _amax_tag(feats)
        taps[-1], pooled = tap_and_pool(feats)             # the last stage's features: a tap AND the global max pool
        taps.append(pooled.unsqueeze(-1).expand(-1, -1, coords.size(-1)))
        return _classify(self.classifier, concat_points(taps, slot, slot_amax))


class _FrustumSegmentation(nn.Module):
    """Foreground / background point segmentation of one frustum (models/kitti/frustum/segmentation/pointnet.py:9-70,
    PVCNN variant): PVConvs at R = 16, 16, 12, 12 -- non-power-of-two grids -- then a SharedMLP to 1024."""
    point_blocks = ((64, 2, 16), (64, 1, 12), (128, 1, 12), (1024, 1, None))

    def __init__(self, num_classes, extra_feature_channels, width_multiplier, voxel_resolution_multiplier):
        super().__init__()
        self.in_channels = extra_feature_channels + 3
        self.num_classes = num_classes
        stages, cin = [], self.in_channels
        for spec in self.point_blocks:
            blocks, cin = _pv_stack(cin, spec, width_multiplier, voxel_resolution_multiplier, with_se=False)
            stages += blocks
        self.point_features = nn.Sequential(*stages)
        self.cloud_features = nn.Sequential()                      # the PVCNN variant has no cloud stages
        layers, _ = _head(cin + cin + num_classes, [512, 256, 128, 128, 0.5, 2], width_multiplier, pointwise=True, classify=True)
        self.classifier = nn.Sequential(*layers)

    def forward(self, inputs):
        feats = inputs['features']
        npts = feats.size(-1)
        # (expand, not repeat: the concatenation reads the broadcast views -- the repeated (B, 3, N) / (B, 1024, N) tensors are never
        # written on their own; reference: .repeat([1, 1, npts]), models/kitti/frustum/segmentation/pointnet.py:52-57, same values)
        one_hot = inputs['one_hot_vectors'].unsqueeze(-1).expand(-1, -1, npts)
        if len(self.cloud_features):                               # (a variant with cloud stages: the modules as the reference chains them)
            per_point, coords = self.point_features((feats, feats[:, :3, :]))
            pooled, _ = self.cloud_features((per_point, coords))
            pooled = pooled.max(dim=-1, keepdim=True).values.expand(-1, -1, npts)
            return _classify(self.classifier, concat_points([one_hot, per_point, pooled]))
        # The PVCNN variant: the last stage's output is BOTH a source of the classifier's concatenation and max-pooled over the points
        # (reference: .max(dim=-1, keepdim=True) on the 134 MB tensor, then torch.cat).  As in workload.PVCNN (round 5): its BatchNorm
        # + ReLU pass emits the row maxima and writes the tensor straight into its slice of the concatenation; the pool's gradient
        # joins the tap's at the winners (_TapAndPool) instead of a dense scatter + a full-tensor addition.
        stages = list(self.point_features)
        coords, x, slot, slot_amax = feats[:, :3, :], feats, None, None
        for i, stage in enumerate(stages):
            if i + 1 == len(stages):
                slot = concat_slot([one_hot], _out_channels(stage), _in_channels(self.classifier), x, self.training)
            with (emit_row_max(_last_norm(stage), out=slot.view if slot is not None else None) if i + 1 == len(stages)
                  else contextlib.nullcontext()):
                x, coords = stage((x, coords))
        if slot is not None and slot.holds(x):
            slot_amax = _amax_tag(x)
        per_point, pooled = tap_and_pool(x)
        pooled = pooled.unsqueeze(-1).expand(-1, -1, npts)
        # (the classifier head module by module on this package's kernels -- under torch.autocast the bare nn.Conv1d at its end would
        # otherwise be a vendor bf16 GEMM with casts either side: _classify; its input from concat_points: one pass, amax table included)
        return _classify(self.classifier, concat_points([one_hot, per_point, pooled], slot, slot_amax))


class _CloudRegressor(nn.Module):
    """SharedMLP stack on coordinates -> global max -> dense head on [descriptor, class one-hot].  With
    `point_attr='features'` / `head_attr='regression'` it is the centre-regression T-Net
    (models/kitti/frustum/center_regression_net.py:9-35); with 'features' / 'classifier' the box-estimation
    PointNet (models/kitti/frustum/box_estimation/pointnet.py:9-52)."""

    def __init__(self, widths, head, num_classes, width_multiplier, head_attr, coords_tuple):
        super().__init__()
        self.in_channels, self.num_classes = 3, num_classes
        self._head_attr, self._coords_tuple = head_attr, coords_tuple
        stack, cin = [], 3
        for w in widths:
            stack.append(SharedMLP(cin, _scaled(w, width_multiplier)))
            cin = _scaled(w, width_multiplier)
        self.features = nn.Sequential(*stack)
        layers, _ = _head(cin + num_classes, head, width_multiplier, pointwise=False, classify=True)
        setattr(self, head_attr, nn.Sequential(*layers))

    def forward(self, inputs):
        coords = inputs['coords']
        if self._coords_tuple:                                     # box estimation feeds (features, coords) tuples
            desc, _ = self.features((coords, coords))
        else:
            desc = self.features(coords)
        desc = desc.max(dim=-1, keepdim=False).values
        # the dense head works on (B, C) numbers: it stays in fp32 under torch.autocast (BASELINE configs[4] asks for bf16 operands in
        # the dense convolutions; three Linear layers on 32 rows gain nothing from them and would cost a cast kernel per operand)
        up = lambda t: t.float() if t.dtype in (torch.bfloat16, torch.float16) else t
        joined = torch.cat([up(desc), up(inputs['one_hot_vectors'])], dim=1)
        with torch.autocast(joined.device.type, enabled=False):
            return run_dense(getattr(self, self._head_attr), joined)     # (the Linear + BatchNorm1d + ReLU blocks: one launch each)


class FrustumPVCNNE(nn.Module):
    """Frustum-PVCNN (efficient variant), BASELINE configs[4]; reference models/kitti/frustum/frustum_net.py:14-76,
    105-113: per-point segmentation with PVConvs -> foreground sampling (logits_mask) -> centre regression ->
    box estimation; returns the reference's dict of box parameters."""

    def __init__(self, num_classes, num_heading_angle_bins, num_size_templates, num_points_per_object, size_templates,
                 extra_feature_channels=1, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        wm = list(width_multiplier) if isinstance(width_multiplier, (list, tuple)) else [width_multiplier] * 3
        self.in_channels = 3 + extra_feature_channels
        self.num_classes = num_classes
        self.num_heading_angle_bins = num_heading_angle_bins
        self.num_size_templates = num_size_templates
        self.num_points_per_object = num_points_per_object
        self.inst_seg_net = _FrustumSegmentation(num_classes, extra_feature_channels, wm[0], voxel_resolution_multiplier)
        self.center_reg_net = _CloudRegressor((128, 128, 256), [256, 128, 3], num_classes, wm[1], 'regression', False)
        n_out = 3 + num_heading_angle_bins * 2 + num_size_templates * 4
        self.box_est_net = _CloudRegressor((128, 128, 256, 512), [512, 256, n_out], num_classes, wm[2], 'classifier', True)
        self.register_buffer('size_templates', size_templates.view(1, num_size_templates, 3))

    def forward(self, inputs):
        import math
        from .modules import functional as PF
        feats, one_hot = inputs['features'], inputs['one_hot_vectors']
        mask_logits = self.inst_seg_net({'features': feats, 'one_hot_vectors': one_hot})
        fg, fg_mean, _ = PF.logits_mask(coords=feats[:, :3, :], logits=mask_logits,
                                        num_points_per_object=self.num_points_per_object)
        delta = self.center_reg_net({'coords': fg, 'one_hot_vectors': one_hot})
        fg = fg - delta.unsqueeze(-1)
        est = self.box_est_net({'coords': fg, 'one_hot_vectors': one_hot})
        nh, ns = self.num_heading_angle_bins, self.num_size_templates
        centre, h_score, h_res, s_score, s_res = est.split([3, nh, nh, ns, ns * 3], dim=-1)
        s_res = s_res.view(-1, ns, 3)
        center_reg = fg_mean + delta
        return {'mask_logits': mask_logits, 'center_reg': center_reg, 'center': centre + center_reg,
                'heading_scores': h_score, 'heading_residuals_normalized': h_res,
                'heading_residuals': h_res * (math.pi / nh), 'size_scores': s_score,
                'size_residuals_normalized': s_res, 'size_residuals': s_res * self.size_templates}


def make_s3dis_batch(batch, num_points, num_classes=13, device='cpu', seed=SEED, duplicates=0.05):
    """Synthetic S3DIS-like batch: features (B,9,N) fp32 and labels (B,N) int64."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(batch, 3, num_points, generator=g) * torch.tensor([1.5, 1.5, 3.0]).view(1, 3, 1)
    rgb = torch.rand(batch, 3, num_points, generator=g)
    room = torch.rand(batch, 3, num_points, generator=g)
    feats = torch.cat([xyz, rgb, room], dim=1)
    ndup = int(num_points * duplicates)
    if ndup:
        src = torch.randint(0, num_points, (batch, ndup), generator=g)
        dst = torch.randint(0, num_points, (batch, ndup), generator=g)
        for b in range(batch):
            feats[b, :, dst[b]] = feats[b, :, src[b]]
    labels = torch.randint(0, num_classes, (batch, num_points), generator=g)
    return feats.contiguous().to(device), labels.to(device)


def make_shapenet_batch(batch, num_points=2048, num_classes=50, num_shapes=16, device='cpu', seed=SEED):
    """Synthetic ShapeNet-part batch (datasets/shapenet.py:62-101): inputs (B, 3+3+16, N) =
    xyz (centred, scaled into the unit ball, jittered by N(0,0.01) clipped to +-0.05), unit normals, the shape
    one-hot repeated over N; points drawn WITH replacement from a smaller pool, so exact duplicates are guaranteed.
    Labels (B,N) int64."""
    g = torch.Generator().manual_seed(seed)
    pool = max(64, num_points * 3 // 4)
    base = torch.randn(batch, 3, pool, generator=g) * torch.tensor([0.35, 0.2, 0.5]).view(1, 3, 1)
    base = base - base.mean(dim=2, keepdim=True)
    base = base / base.norm(dim=1, keepdim=True).amax(dim=2, keepdim=True)          # unit ball
    pick = torch.randint(0, pool, (batch, num_points), generator=g)
    xyz = torch.gather(base, 2, pick.unsqueeze(1).expand(-1, 3, -1))
    xyz = xyz + (torch.randn(batch, 3, num_points, generator=g) * 0.01).clamp(-0.05, 0.05)
    # keep ~6 % exact duplicates un-jittered (the loader's replace=True draws)
    ndup = num_points // 16
    src = torch.randint(0, num_points, (batch, ndup), generator=g)
    dst = torch.randint(0, num_points, (batch, ndup), generator=g)
    for b in range(batch):
        xyz[b, :, dst[b]] = xyz[b, :, src[b]]
    normals = torch.randn(batch, 3, num_points, generator=g)
    normals = normals / normals.norm(dim=1, keepdim=True).clamp(min=1e-6)
    shape_id = torch.randint(0, num_shapes, (batch,), generator=g)
    one_hot = torch.zeros(batch, num_shapes, num_points)
    one_hot[torch.arange(batch), shape_id, :] = 1.0
    labels = torch.randint(0, num_classes, (batch, num_points), generator=g)
    return torch.cat([xyz, normals, one_hot], dim=1).contiguous().to(device), labels.to(device)


def make_frustum_batch(batch, num_points=1024, num_classes=3, device='cpu', seed=SEED):
    """Synthetic KITTI frustum batch: {'features': (B,4,N) = frustum-like xyz (x, y ~ N(0,1.5),
    depth z ~ U[5,40]) + intensity U[0,1], 'one_hot_vectors': (B,3)}, and per-point foreground labels (B,N)."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.randn(batch, 2, num_points, generator=g) * 1.5
    z = torch.rand(batch, 1, num_points, generator=g) * 35.0 + 5.0
    inten = torch.rand(batch, 1, num_points, generator=g)
    cls = torch.randint(0, num_classes, (batch,), generator=g)
    one_hot = torch.zeros(batch, num_classes)
    one_hot[torch.arange(batch), cls] = 1.0
    labels = torch.randint(0, 2, (batch, num_points), generator=g)
    return ({'features': torch.cat([xy, z, inten], dim=1).contiguous().to(device), 'one_hot_vectors': one_hot.to(device)},
            labels.to(device))


def frustum_size_templates(num_size_templates=8):
    """Mean (l, w, h) box sizes per class; values only matter as a fixed buffer of the right shape here
    (the reference reads them from its KITTI attributes, datasets/kitti/attributes.py)."""
    base = torch.tensor([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73], [5.06, 1.9, 2.2],
                         [10.1, 2.6, 3.0], [2.1, 1.2, 1.5], [16.2, 2.6, 3.5], [0.84, 0.66, 1.76]])
    return base[:num_size_templates].clone()


def make_frustum_targets(batch, num_points=1024, num_heading_angle_bins=12, num_size_templates=8, device='cpu', seed=SEED):
    """Synthetic targets for FrustumPointNetLoss (modules/frustum.py:43-124): foreground labels, box centre, heading bin +
    residual, size template + residual."""
    import math
    g = torch.Generator().manual_seed(seed + 17)
    t = {'mask_logits': torch.randint(0, 2, (batch, num_points), generator=g),
         'center': torch.randn(batch, 3, generator=g) * 2 + torch.tensor([0.0, 0.0, 20.0]),
         'heading_bin_id': torch.randint(0, num_heading_angle_bins, (batch,), generator=g),
         'heading_residual': (torch.rand(batch, generator=g) - 0.5) * (2 * math.pi / num_heading_angle_bins),
         'size_template_id': torch.randint(0, num_size_templates, (batch,), generator=g),
         'size_residual': torch.randn(batch, 3, generator=g) * 0.1}
    return {k: v.to(device) for k, v in t.items()}
