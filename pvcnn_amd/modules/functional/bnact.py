"""batch_norm_act: BatchNorm{1,2,3}d fused with the ReLU / LeakyReLU that follows it.

The reference stacks them as separate modules (modules/pvconv.py:20-27, modules/shared_mlp.py:20-25).
Semantics kept: training mode normalises with biased batch statistics and updates running_mean /
running_var (unbiased variance, `momentum`; cumulative average when momentum is None) and
num_batches_tracked exactly like torch.nn.BatchNorm; eval mode uses the running statistics.
`run_layers` walks an nn.Sequential and fuses every (BatchNorm, activation) pair it meets on a GPU
tensor -- parameters, buffers and state_dict keys stay those of the plain modules."""
import contextlib
import threading

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _cache, _gradslots
from ._autograd import native, amp_fwd, amp_bwd
from .devoxelization import CornerTaps


def _rows(t, shape):
    """grad tensor -> (B, C, S) view without a copy when each sample's rows are contiguous (channel slices of a
    wider tensor, as torch.cat's backward produces), else a contiguous copy."""
    from .backend import batch_strided_ok
    if t.dim() == len(shape) and t.dim() >= 3:
        try:
            v = t.view(shape[0], shape[1], -1) if t.dim() > 3 else t
            if batch_strided_ok(v):
                return v
        except RuntimeError:
            pass
    return t.contiguous().view(shape[0], shape[1], -1)


def _servable(x):
    """The GPU path's kernels take fp32 tensors; under torch.autocast a half / bfloat16 tensor produced by a torch op in between is
    cast up at the op boundary (custom_fwd(cast_inputs=float32) of every autograd node here), so it is served as well."""
    return x.dtype == torch.float32 or (torch.is_autocast_enabled() and x.dtype in (torch.bfloat16, torch.float16))


def _amax_seg_for(shape, is_cuda):
    """Segment length of the amax buffer (include/pvcnn_hip.h) the convolution NEXT TO a BatchNorm over a tensor of `shape` wants of
    that tensor, or 0: a cubic voxel grid (B,C,R,R,R) feeds / is fed by a 3x3x3 convolution in f16x2 arithmetic -> one z row (R);
    point features (B,C,N) / (B,C,M,U) a 1x1 GEMM -> its 256-point tile.  0 when that arithmetic is off (no table is needed).
    Under bf16 autocast the forward / backward-data products need no scales, but the backward-weight kernels are the f16x2 ones and take
    word [0] of the same buffers (else: one global-maximum pass + memset per operand and layer, 0.4 ms per Frustum-PVCNN step)."""
    be = native()
    if not is_cuda:
        return 0
    cap = getattr(be, 'BNACT_AMAX_MAX_SEG', 0)
    if len(shape) == 5 and shape[2] == shape[3] == shape[4]:
        ok = getattr(be, 'has_conv3d_split', False) and getattr(be, 'conv_math', '') == 'f16x2'
        return int(shape[2]) if ok and shape[2] <= cap else 0
    if len(shape) in (3, 4):
        ok = getattr(be, 'has_pwconv_split', False) and getattr(be, 'pw_math', '') == 'f16x2'
        seg = getattr(be, 'PW_AMAX_SEG', 0)
        return seg if ok and 0 < seg <= cap else 0
    return 0


def _bnact_backward(x3, g3, w, b, mean, rstd, slope, training, shape, drop=None):
    """native bnact_backward -> (grad_x viewed as `shape`, grad_gamma, grad_beta); the apply pass also leaves grad_x's amax buffer
    on the returned tensor (_cache.tag_amax) for the f16x2 backward products of the convolution in front of this BatchNorm."""
    be = native()
    seg = _amax_seg_for(shape, x3.is_cuda)
    dst = _gradslots.destinations(be, w, b)       # gamma's / beta's slots in a flat gradient bucket, where there is one
    if seg:
        gx, gw, gb, amax = be.bnact_backward(x3, g3, w, b, mean, rstd, slope, training, amax_seg=seg, **({'drop': drop} if drop else {}), **dst)
        return _cache.tag_amax(gx.view(shape), seg, amax), gw, gb
    assert drop is None, 'the fused dropout rides on the amax-emitting passes'  
    gx, gw, gb = be.bnact_backward(x3, g3, w, b, mean, rstd, slope, training, **dst)
    return gx.view(shape), gw, gb


__all__ = ['batch_norm_act', 'batch_norm_act_devoxelize', 'batch_norm_act_se_devoxelize', 'fusable_tail', 'run_layers', 'fused_dropout_ok']


class BatchNormAct(Function):
    @staticmethod
    @amp_fwd
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope, stats_part=None, stats_shift=None,
                amax_seg=0, counter=None, drop_p=0.0, drop_seed=None, row_max=False):
        """-> y, or (y, y's amax buffer) when amax_seg > 0 (second output: not differentiable).
        drop_p > 0 (with amax_seg > 0): y = dropout(act(bn(x)), drop_p) with the keep decisions of csrc/bnact.hip under drop_seed.
        row_max (amax_seg > 0, statistics from an epilogue, no dropout): -> (y, amax, winners, values), the last two == y.max(dim=-1)'s
        indices and values, emitted by the apply pass (emit_row_max)."""
        shape = x.shape
        x3 = x.contiguous().view(shape[0], shape[1], -1)
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        stats = armed = None
        if training and stats_part is not None:   # partial sums from the producing convolution's epilogue
            if amax_seg and row_max:              # ... and zeroes the row keys of the max-pool that follows, in the same buffer
                whole, armed, keys = native().amax_and_row_keys(x3.shape[0], x3.shape[1], x3.shape[2], amax_seg, x3.device)
                stats = native().bn_finalize(stats_part, x3.shape[0] * x3.shape[2], running_mean, running_var, momentum, eps, stats_shift,
                                             zero_word=whole, counter=counter)
                ctx.slope, ctx.training, ctx.shape, ctx.drop = slope, training, shape, None
                dest = getattr(_ROW_MAX, 'out', None)       # (emit_row_max(bn, out=...): y goes straight into its slice of the concatenation)
                if dest is not None and not (dest.is_cuda and tuple(dest.shape) == tuple(x3.shape) and dest.dtype == torch.float32):
                    dest = None
                y, winners, values = native().bnact_apply_rowmax(x3, w, b, stats[0], stats[1], slope, amax_seg, armed, keys, out=dest)
                ctx.save_for_backward(x3, w, b, stats[0], stats[1])
                ctx.mark_non_differentiable(armed, winners, values)
                ctx.set_materialize_grads(False)
                return y.view(shape), armed, winners, values
            if amax_seg:                          # the finalize launch also arms word [0] of the amax buffer the apply pass fills
                armed = native().amax_buffer(x3.shape[0], x3.shape[2], amax_seg, x3.device)
                stats = native().bn_finalize(stats_part, x3.shape[0] * x3.shape[2], running_mean, running_var, momentum, eps, stats_shift,
                                             zero_word=armed, counter=counter)
            else:
                stats = native().bn_finalize(stats_part, x3.shape[0] * x3.shape[2], running_mean, running_var, momentum, eps, stats_shift,
                                             counter=counter)
        ctx.slope, ctx.training, ctx.shape = slope, training, shape
        ctx.drop = (drop_seed, float(drop_p)) if (drop_p and drop_seed is not None) else None
        if amax_seg:
            y, mean, rstd, amax = native().bnact_forward(x3, w, b, running_mean, running_var, training, momentum, eps, slope, stats=stats,
                                                         amax_seg=amax_seg, y_amax=armed, **({'drop': ctx.drop} if ctx.drop else {}))
            ctx.save_for_backward(x3, w, b, mean, rstd)
            ctx.mark_non_differentiable(amax)
            ctx.set_materialize_grads(False)
            return y.view(shape), amax
        y, mean, rstd = native().bnact_forward(x3, w, b, running_mean, running_var, training, momentum, eps, slope, stats=stats)
        ctx.save_for_backward(x3, w, b, mean, rstd)
        return y.view(shape)

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_y, grad_amax=None, grad_winners=None, grad_values=None):
        if grad_y is None:
            return (None,) * 16
        x3, w, b, mean, rstd = ctx.saved_tensors
        g3 = _rows(grad_y, ctx.shape)
        gx, gw, gb = _bnact_backward(x3, g3, w, b, mean, rstd, ctx.slope, ctx.training, ctx.shape, drop=ctx.drop)
        return (gx, gw if w is not None else None, gb if b is not None else None,
                None, None, None, None, None, None, None, None, None, None, None, None, None)


# ---- the max-pool over the points behind a (BatchNorm, activation) pair, emitted by the pair's apply pass ----------------------------
_ROW_MAX = threading.local()


@contextlib.contextmanager
def emit_row_max(bn, out=None):
    """with emit_row_max(bn): inside, the fused (BatchNorm `bn`, activation) pass ALSO emits the row maxima of the (B, C, N) tensor it
    writes; the tensor then carries them as `_pvcnn_row_max` = (winners (B,C) int64, values (B,C)) == y.max(dim=-1)'s (indices, values)
    -- what the global max-pool of models/s3dis/pvcnn.py:41-43 needs, without a read of the tensor.  Only the training path whose
    statistics come from the producing convolution's epilogue takes it; elsewhere nothing is attached and the caller reduces itself.
    out (round 5): a (B, C, N) view -- the channel slice of the classifier's concatenation (models/s3dis/pvcnn.py:45) this tensor will
    be -- that the same pass writes INSTEAD of a tensor of its own: `workload.concat_points(..., out=)` then copies nothing for it (the
    widest source: 268 of PVCNN's 386 MB).  Ignored wherever the row-maxima pass itself is not taken."""
    prev = (getattr(_ROW_MAX, 'bn', None), getattr(_ROW_MAX, 'out', None))
    _ROW_MAX.bn, _ROW_MAX.out = bn, out
    try:
        yield
    finally:
        _ROW_MAX.bn, _ROW_MAX.out = prev


def _bn_mode(bn, finalize_counts=False):
    """-> (use_batch_stats, momentum, running_mean, running_var, counter) of one forward call of module `bn`, with the side effects
    torch.nn.BatchNorm has (num_batches_tracked).  finalize_counts: the statistics of this call come from a convolution epilogue and
    are finalised by a kernel of ours -- that launch also increments num_batches_tracked (returned as `counter`; otherwise None and
    the counter was incremented here, one tiny launch per BatchNorm and step)."""
    use_batch_stats = bn.training or bn.running_mean is None
    momentum = 0.0 if bn.momentum is None else bn.momentum
    counter = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        nbt = bn.num_batches_tracked
        if (finalize_counts and bn.momentum is not None and nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1):
            counter = nbt
        else:
            nbt.add_(1)
            if bn.momentum is None:                  # cumulative moving average
                momentum = 1.0 / float(nbt)
    rm = bn.running_mean if (bn.track_running_stats or not use_batch_stats) else None
    rv = bn.running_var if (bn.track_running_stats or not use_batch_stats) else None
    return use_batch_stats, momentum, rm, rv, counter


def _split(stats):
    """stats: None | (partial sums, shift) as produced by run_layers -> (part, shift)."""
    if stats is None:
        return None, None
    part, shift = stats
    return part, (shift.detach().contiguous() if shift is not None else None)


def fused_dropout_ok(x):
    """Can an nn.Dropout in training mode behind a (BatchNorm, activation) pair over x ride on the pair's passes (batch_norm_act(...,
    drop_p=p))?  Needs the amax-emitting passes (csrc/bnact.hip) -- i.e. the f16x2 arithmetic of the convolution next door -- and no
    autocast."""
    return (x.is_cuda and x.dtype == torch.float32 and getattr(native(), 'has_bnact_dropout', False) and not torch.is_autocast_enabled()
            and _amax_seg_for(x.shape, True) > 0 and 0 < x.numel() < (1 << 33))


def batch_norm_act(x, bn, slope, stats_part=None, drop_p=0.0):
    """Apply BatchNorm module `bn` followed by LeakyReLU(slope) (slope = 0: ReLU) to x (B, C, ...).
    stats_part: (per-workgroup partial sums of x - shift written by the convolution that produced x, shift = its bias).
    drop_p > 0 (only where fused_dropout_ok(x)): ... followed by a training-mode nn.Dropout(drop_p), in the same passes; the keep
    decisions are a function of one int64 drawn here from torch's generator (so torch.manual_seed governs them, and a captured graph
    draws a fresh one per replay) -- a different stream than torch.nn.functional.dropout's, the same distribution."""
    use_batch_stats, momentum, rm, rv, counter = _bn_mode(bn, finalize_counts=stats_part is not None and bn.training)
    part, shift = _split(stats_part)
    # the apply pass emits the f16x2 scale table of what it writes for the convolution that (usually) consumes it
    seg = _amax_seg_for(x.shape, x.is_cuda)
    if (seg and bn is getattr(_ROW_MAX, 'bn', None) and not drop_p and use_batch_stats and part is not None and x.dim() == 3
            and x.shape[2] % 256 == 0 and seg % 4 == 0 and x.dtype == torch.float32 and x.data_ptr() % 16 == 0 and x.is_contiguous()
            and getattr(native(), 'has_bnact_rowmax', False)):
        y, amax, winners, values = BatchNormAct.apply(x, bn.weight, bn.bias, rm, rv, use_batch_stats, momentum, bn.eps, slope, part, shift,
                                                      seg, counter, 0.0, None, True)
        y._pvcnn_row_max = (winners, values, y._version)     # (keyed by the in-place version: workload.tap_and_pool re-checks it)
        return _cache.tag_amax(y, seg, amax)
    if seg:
        seed = torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64, device=x.device) if drop_p else None
        y, amax = BatchNormAct.apply(x, bn.weight, bn.bias, rm, rv, use_batch_stats, momentum, bn.eps, slope, part, shift, seg, counter,
                                     float(drop_p), seed)
        return _cache.tag_amax(y, seg, amax)
    assert not drop_p, 'fused dropout: see fused_dropout_ok'
    return BatchNormAct.apply(x, bn.weight, bn.bias, rm, rv, use_batch_stats, momentum, bn.eps, slope, part, shift, 0, counter)


class BatchNormActDevoxelize(Function):
    """trilinear_devoxelize(leaky_relu(batch_norm(grid)), coords): PVConv's last BatchNorm3d + LeakyReLU and the
    devoxelization (modules/pvconv.py:25-27,36) with the activated grid never written to memory -- the gather
    kernel normalises and activates while it stages the grid into LDS.  Bit-identical to the two separate ops."""

    @staticmethod
    @amp_fwd
    def forward(ctx, grid, coords, weight, bias, running_mean, running_var, use_batch_stats, momentum, eps, slope,
                resolution, is_training, stats_part=None, stats_shift=None, addend=None, counter=None):
        shape = grid.shape
        x3 = grid.contiguous().view(shape[0], shape[1], -1)
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        if use_batch_stats and stats_part is not None:
            mean, rstd = native().bn_finalize(stats_part, x3.shape[0] * x3.shape[2], running_mean, running_var, momentum, eps, stats_shift,
                                              counter=counter)
        elif use_batch_stats:
            mean, rstd = native().bn_stats(x3, running_mean, running_var, momentum, eps)
        else:
            mean, rstd = running_mean.contiguous(), torch.rsqrt(running_var + eps)
        r = int(resolution)
        pts = coords.contiguous()
        add = addend.contiguous() if addend is not None else None      # the point branch: added in the gather's store
        ctx.has_addend = add is not None
        if not is_training:
            return native().trilinear_devoxelize_bnact_forward(r, False, pts, x3, w, b, mean, rstd, slope, add)[0]
        taps = CornerTaps.of(coords, r)
        out = taps.forward(lambda emit: native().trilinear_devoxelize_bnact_forward(r, emit, pts, x3, w, b, mean, rstd, slope, add))
        ctx.save_for_backward(x3, w, b, mean, rstd, taps.inds, taps.wgts)
        ctx.taps = taps
        ctx.slope, ctx.use_batch_stats, ctx.shape = slope, use_batch_stats, shape
        return out

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_out):
        x3, w, b, mean, rstd, _, _ = ctx.saved_tensors
        g_act = ctx.taps.backward(_rows(grad_out, grad_out.shape))
        gx, gw, gb = _bnact_backward(x3, g_act.view(x3.shape), w, b, mean, rstd, ctx.slope, ctx.use_batch_stats, ctx.shape)
        return (gx, None, gw if w is not None else None, gb if b is not None else None,
                None, None, None, None, None, None, None, None, None, None, grad_out if ctx.has_addend else None, None)


def batch_norm_act_devoxelize(grid, coords, bn, slope, resolution, is_training, stats_part=None, addend=None):
    """trilinear_devoxelize(act(bn(grid)), coords) [+ addend]: PVConv's tail (modules/pvconv.py:25-27,36-38) in one gather."""
    use_batch_stats, momentum, rm, rv, counter = _bn_mode(bn, finalize_counts=stats_part is not None and bn.training)
    part, shift = _split(stats_part)
    return BatchNormActDevoxelize.apply(grid, coords, bn.weight, bn.bias, rm, rv, use_batch_stats, momentum, bn.eps, slope,
                                        resolution, is_training, part, shift, addend, counter)


class BatchNormActSEDevoxelize(Function):
    """trilinear_devoxelize(SE3d(leaky_relu(batch_norm(grid))), coords) [+ addend]: the tail of a PVConv WITH squeeze-and-excitation
    (modules/pvconv.py:25-30,36-38, modules/se.py:6-17) as ONE node that touches the convolution's output three times in all:

      forward : one reduction pass over the grid -> per (cloud, channel) A = sum act'(z), Ax = sum act'(z) * xhat (z = gamma * xhat +
                beta, xhat = (x - mean) * rstd).  LeakyReLU is piecewise linear through 0, act(z) = z * act'(z), so the squeeze is
                mean act(z) = (gamma * Ax + beta * A) / S -- no activated grid is written for it; the two tiny Linear layers + sigmoid
                run on (B, C) numbers; the excitation rides on the gather's row transform (a second rounded multiplication, as in
                the reference).  Neither act(bn(x)) nor its product with the excitation ever exists in memory.
      backward: the devoxelize scatter gives g_y = dL/d(a * s); ONE reduction pass yields P = sum g_y act', Q = sum g_y act' xhat per
                (cloud, channel): dL/ds = sum g_y a = gamma * Q + beta * P feeds the excitation's backward (tiny), whose dL/dmean
                comes back as a per-(cloud, channel) constant gm; the BatchNorm sums follow WITHOUT another pass,
                sum g' = s * P + gm * A, sum g' xhat = s * Q + gm * Ax with g' = (s * g_y + gm) * act'(z); one apply pass writes grad_x.
    The reference's graph (activated grid, mean, product, devoxelize: 3 writes + 4 reads of the grid forward, as many backward)
    computes the same numbers; summation order differs (<= 1e-6 relative), the discrete decisions (act' = sign of z) are the same."""

    @staticmethod
    @amp_fwd
    def forward(ctx, grid, coords, weight, bias, running_mean, running_var, use_batch_stats, momentum, eps, slope,
                resolution, is_training, stats_part, stats_shift, addend, counter, fc1, fc2):
        be = native()
        shape = grid.shape
        x3 = grid.contiguous().view(shape[0], shape[1], -1)
        nb, nc, s3 = x3.shape
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        if use_batch_stats and stats_part is not None:
            mean, rstd = be.bn_finalize(stats_part, nb * s3, running_mean, running_var, momentum, eps, stats_shift, counter=counter)
        elif use_batch_stats:
            mean, rstd = be.bn_stats(x3, running_mean, running_var, momentum, eps)
        else:
            mean, rstd = running_mean.contiguous(), torch.rsqrt(running_var + eps)
        # squeeze from the two sums (grad_y == 1 in the reduction kernel of the BatchNorm backward)
        w1, w2 = fc1.contiguous(), fc2.contiguous()
        fused_se = getattr(be, 'has_se_excite', False) and nc <= 2048 and w1.shape[0] <= 256
        if fused_se:     # the slice sums of the pass + the excitation: one launch (csrc/se.hip)
            part = be.bnact_partial_sums_raw(x3, None, w, b, mean, rstd, slope)
            a_sum, ax_sum, squeezed, hidden, excite = be.se_excite_forward(part, w, b, w1, w2, s3)
        else:
            a_sum, ax_sum = be.bnact_partial_sums(x3, None, w, b, mean, rstd, slope)        # (B, C) each
            gam = w if w is not None else torch.ones_like(mean)
            bet = b if b is not None else torch.zeros_like(mean)
            squeezed = (gam * ax_sum + bet * a_sum) / float(s3)
            hidden = torch.relu(squeezed @ w1.t())
            excite = torch.sigmoid(hidden @ w2.t()).contiguous()                             # (B, C)
        r = int(resolution)
        pts = coords.contiguous()
        add = addend.contiguous() if addend is not None else None
        ctx.has_addend = add is not None
        if not is_training:
            return be.trilinear_devoxelize_bnact_forward(r, False, pts, x3, w, b, mean, rstd, slope, add, se_scale=excite)[0]
        taps = CornerTaps.of(coords, r)
        out = taps.forward(lambda emit: be.trilinear_devoxelize_bnact_forward(r, emit, pts, x3, w, b, mean, rstd, slope, add, se_scale=excite))
        ctx.save_for_backward(x3, w, b, mean, rstd, a_sum, ax_sum, squeezed, hidden, excite, w1, w2, taps.inds, taps.wgts)
        ctx.taps = taps
        ctx.slope, ctx.use_batch_stats, ctx.shape = slope, use_batch_stats, shape
        return out

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_out):
        be = native()
        x3, w, b, mean, rstd, a_sum, ax_sum, squeezed, hidden, excite, w1, w2, _, _ = ctx.saved_tensors
        nb, nc, s3 = x3.shape
        g_y = ctx.taps.backward(_rows(grad_out, grad_out.shape)).view(x3.shape)             # dL/d(act(bn(x)) * excite)
        if getattr(be, 'has_se_excite', False) and nc <= 2048 and w1.shape[0] <= 256:
            # the slice sums P, Q + the excitation backward + the BatchNorm sums below in two launches (csrc/se.hip)
            part = be.bnact_partial_sums_raw(x3, g_y, w, b, mean, rstd, ctx.slope)
            g_w1, g_w2, g_mean, sum_beta, sum_gamma = be.se_excite_backward(part, a_sum, ax_sum, w, b, squeezed, hidden, excite, w1, w2, s3)
        else:
            p_sum, q_sum = be.bnact_partial_sums(x3, g_y, w, b, mean, rstd, ctx.slope)      # (B, C) each
            gam = w if w is not None else torch.ones_like(mean)
            bet = b if b is not None else torch.zeros_like(mean)
            # excitation backward: s = sigmoid(relu(m W1^T) W2^T)
            g_excite = gam * q_sum + bet * p_sum
            g_pre2 = g_excite * excite * (1.0 - excite)
            g_w2 = g_pre2.t() @ hidden
            g_pre1 = (g_pre2 @ w2) * (hidden > 0).to(hidden.dtype)
            g_w1 = g_pre1.t() @ squeezed
            g_mean = ((g_pre1 @ w1) / float(s3)).contiguous()                                # dL/d(squeezed) spread over the S voxels
            # BatchNorm sums of g' = (excite * g_y + g_mean) * act'(z), from the four per-(cloud, channel) sums
            sum_beta = (excite * p_sum + g_mean * a_sum).sum(dim=0).contiguous()
            sum_gamma = (excite * q_sum + g_mean * ax_sum).sum(dim=0).contiguous()
        seg = _amax_seg_for(ctx.shape, x3.is_cuda) or 256
        gx, amax = be.bnact_backward_apply(x3, g_y, w, b, mean, rstd, sum_gamma, sum_beta, ctx.slope, ctx.use_batch_stats,
                                           bc_mul=excite, bc_add=g_mean, amax_seg=seg)
        gx = gx.view(ctx.shape)
        if _amax_seg_for(ctx.shape, x3.is_cuda):
            _cache.tag_amax(gx, seg, amax)
        return (gx, None, sum_gamma if w is not None else None, sum_beta if b is not None else None,
                None, None, None, None, None, None, None, None, None, None, grad_out if ctx.has_addend else None, None, g_w1, g_w2)


def batch_norm_act_se_devoxelize(grid, coords, bn, slope, se, resolution, is_training, stats_part=None, addend=None):
    """trilinear_devoxelize(se(act(bn(grid))), coords) [+ addend]: PVConv's tail with squeeze-and-excitation in one node."""
    use_batch_stats, momentum, rm, rv, counter = _bn_mode(bn, finalize_counts=stats_part is not None and bn.training)
    part, shift = _split(stats_part)
    return BatchNormActSEDevoxelize.apply(grid, coords, bn.weight, bn.bias, rm, rv, use_batch_stats, momentum, bn.eps, slope,
                                          resolution, is_training, part, shift, addend, counter, se.fc[0].weight, se.fc[2].weight)


def _plain_se(m):
    """True for the reference's SE3d layout (modules/se.py:9-14): Linear (no bias), ReLU, Linear (no bias), Sigmoid."""
    fc = getattr(m, 'fc', None)
    return (isinstance(fc, nn.Sequential) and len(fc) == 4 and isinstance(fc[0], nn.Linear) and fc[0].bias is None
            and isinstance(fc[1], nn.ReLU) and isinstance(fc[2], nn.Linear) and fc[2].bias is None and isinstance(fc[3], nn.Sigmoid)
            and not _has_hooks(fc) and not any(_has_hooks(f) for f in fc))


def fusable_tail(layers, x):
    """If the nn.Sequential ends in (BatchNorm, ReLU|LeakyReLU [, SE3d]) and the GPU path can fuse that tail into the
    devoxelize gather for tensor x (the Sequential's INPUT: same device / dtype), return (bn, slope, se | None)."""
    mods = list(layers)
    if _has_hooks(layers) or any(_has_hooks(m) for m in mods):   # hooks only fire through __call__: run module by module then
        return None
    be = native()
    if not (x.is_cuda and _servable(x) and getattr(be, 'has_devox_bnact', False)):
        return None
    if (len(mods) >= 3 and _plain_se(mods[-1]) and getattr(be, 'has_bnact_split_bwd', False)
            and isinstance(mods[-3], nn.modules.batchnorm._BatchNorm) and _slope(mods[-2]) is not None):
        return mods[-3], _slope(mods[-2]), mods[-1]
    if len(mods) >= 2 and isinstance(mods[-2], nn.modules.batchnorm._BatchNorm) and _slope(mods[-1]) is not None:
        return mods[-2], _slope(mods[-1]), None
    return None


def _slope(act):
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    return None


def _has_hooks(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None))


def _is_pointwise(m):
    """kernel-size-1, stride-1, unpadded, ungrouped Conv1d / Conv2d: the SharedMLP convolutions."""
    return (isinstance(m, (nn.Conv1d, nn.Conv2d)) and all(k == 1 for k in m.kernel_size) and all(v == 1 for v in m.stride)
            and all(v == 0 for v in m.padding) and all(v == 1 for v in m.dilation) and m.groups == 1
            and isinstance(m.padding, tuple))


def _wants_batch_stats(m):
    return isinstance(m, nn.modules.batchnorm._BatchNorm) and (m.training or m.running_mean is None)


def run_layers(layers, x, stop=None, tail_stats=False, tail_dropout=0.0):
    """nn.Sequential.forward with the GPU path's own kernels: 1x1 convolutions as channel-major MFMA GEMMs,
    (BatchNorm, ReLU|LeakyReLU) pairs fused, and the statistics of a BatchNorm that directly follows one of our
    convolutions taken from that convolution's epilogue instead of a pass over its output.
    `stop`: run only the first `stop` modules.  `tail_stats`: return (x, stats_part) where stats_part belongs
    to the BatchNorm at position `stop` if the last module run was such a convolution (else None).
    `tail_dropout` = p > 0: a training-mode nn.Dropout(p) follows the stack and its LAST two modules are a fusable (BatchNorm,
    activation) pair: the pair's passes apply it (the caller has checked fused_dropout_ok and skips the Dropout module)."""
    all_mods = list(layers)
    mods = all_mods[:stop]
    # forward / backward hooks on a sub-module (FLOP counters, feature extractors, pruning) only fire through its
    # __call__: a hooked Sequential runs module by module like nn.Sequential does
    hooked = any(_has_hooks(m) for m in all_mods)
    fuse = x.is_cuda and getattr(native(), 'has_bnact', False) and not hooked
    pw = x.is_cuda and getattr(native(), 'has_pwconv', False) and not hooked
    part = carried = None
    dropped = False
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = all_mods[i + 1] if i + 1 < len(all_mods) else None
        want = fuse and _servable(x) and _wants_batch_stats(nxt)
        part = None
        if pw and _is_pointwise(m) and _servable(x) and x.dim() == len(m.kernel_size) + 2 and x.numel() > 0:
            from .pwconv import pointwise_conv, pw_nsplit
            split = pw_nsplit(x, m.weight)        # decided HERE: inside the autograd node autocast is already switched off
            if want:
                x, part = pointwise_conv(x, m.weight, m.bias, True, split)
                part = (part, m.bias)
            else:
                x = pointwise_conv(x, m.weight, m.bias, False, split)
            i += 1
        elif want and hasattr(m, 'forward_with_stats') and x.numel() > 0 and not hooked:
            res = m.forward_with_stats(x)
            if isinstance(res, tuple):
                x, part = res
                part = (part, m.bias)
            else:
                x = res
            i += 1
        elif (fuse and isinstance(m, nn.modules.batchnorm._BatchNorm) and i + 1 < len(mods) and x.dim() >= 3
                and _servable(x) and _slope(mods[i + 1]) is not None and x.numel() > 0):
            drop_here = tail_dropout if (tail_dropout and i + 2 == len(mods) and fused_dropout_ok(x)) else 0.0
            x = batch_norm_act(x, m, _slope(mods[i + 1]), stats_part=carried, drop_p=drop_here)
            dropped = dropped or bool(drop_here)
            i += 2
        else:
            x = m(x)
            i += 1
        carried = part
    if tail_dropout and not dropped:            # the last pair was not fusable after all: the Dropout the caller skipped, as torch runs it
        x = torch.nn.functional.dropout(x, float(tail_dropout), True)
    return (x, part) if tail_stats else x
