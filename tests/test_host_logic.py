"""Host-side logic of the module layer that needs no GPU: layer dispatch, stride handling, batch sharding."""
import pytest
import torch
import torch.nn as nn


def test_batch_strided_gradient_views_are_recognised():
    from pvcnn_amd.modules.functional.backend import batch_strided_ok
    from pvcnn_amd.modules.functional.bnact import _rows
    wide = torch.randn(3, 40, 16)
    sl = wide[:, 8:24, :]                                   # what torch.cat's backward hands to a consumer
    assert not sl.is_contiguous() and batch_strided_ok(sl)
    v = _rows(sl, sl.shape)
    assert v.data_ptr() == sl.data_ptr() and v.stride(0) == 40 * 16, 'channel slices must be consumed in place'
    assert batch_strided_ok(torch.randn(2, 5, 7))
    assert not batch_strided_ok(wide.transpose(1, 2))       # rows not contiguous: needs a copy
    assert not batch_strided_ok(wide[:, :, ::2])
    assert not batch_strided_ok(torch.randn(2, 5, 7).double())
    t = _rows(wide.transpose(1, 2), (3, 16, 40))
    assert t.is_contiguous() and t.shape == (3, 16, 40)
    g5 = torch.randn(2, 6, 4, 4, 4)[:, 1:5]                 # 5-D slice: viewable as (B, C, S) without a copy
    v5 = _rows(g5, g5.shape)
    assert v5.shape == (2, 4, 64) and v5.data_ptr() == g5.data_ptr()


def test_pointwise_detection():
    from pvcnn_amd.modules.functional.bnact import _is_pointwise
    assert _is_pointwise(nn.Conv1d(8, 4, 1))
    assert _is_pointwise(nn.Conv2d(8, 4, 1))
    assert _is_pointwise(nn.Conv2d(8, 4, (1, 1), bias=False))
    assert not _is_pointwise(nn.Conv1d(8, 4, 3, padding=1))
    assert not _is_pointwise(nn.Conv1d(8, 4, 1, stride=2))
    assert not _is_pointwise(nn.Conv1d(8, 8, 1, groups=2))
    assert not _is_pointwise(nn.Conv3d(8, 4, 1))
    assert not _is_pointwise(nn.Linear(8, 4))


def test_run_layers_on_cpu_is_the_plain_sequential():
    """CPU tensors never reach a native kernel: run_layers must be nn.Sequential.forward (same bits, same buffers)."""
    from pvcnn_amd.modules.functional.bnact import fusable_tail, run_layers
    torch.manual_seed(0)
    seq = nn.Sequential(nn.Conv1d(6, 8, 1), nn.BatchNorm1d(8), nn.ReLU(True), nn.Conv1d(8, 5, 1), nn.BatchNorm1d(5), nn.ReLU(True)).train()
    ref = nn.Sequential(*list(seq))                                            # same module objects
    x = torch.randn(4, 6, 32)
    state = {k: v.clone() for k, v in seq.state_dict().items()}
    a = run_layers(seq, x)
    seq.load_state_dict(state)
    b = ref(x)
    assert torch.equal(a, b)
    seq.load_state_dict(state)
    y, part = run_layers(seq, x, stop=4, tail_stats=True)
    assert part is None and y.shape == (4, 5, 32)
    assert fusable_tail(seq, x) is None                       # the fused tail is a GPU-only path


def test_voxelization_cpu_formula_matches_the_reference_expression():
    from pvcnn_amd.modules import Voxelization
    torch.manual_seed(1)
    coords = torch.randn(2, 3, 50)
    for normalize, eps in [(True, 0.0), (True, 1e-3), (False, 0.0)]:
        v = Voxelization(8, normalize=normalize, eps=eps)
        got = v.normalized_coords(coords)
        c = coords - coords.mean(2, keepdim=True)
        want = c / (c.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values * 2.0 + eps) + 0.5 if normalize else (c + 1) / 2.0
        assert torch.equal(got, torch.clamp(want * 8, 0, 7))


def test_shard_batch():
    from pvcnn_amd.dp import shard_batch
    assert [shard_batch(16, 4, r) for r in range(4)] == [slice(0, 4), slice(4, 8), slice(8, 12), slice(12, 16)]
    with pytest.raises(ValueError):
        shard_batch(10, 4, 0)


def test_coordinate_memo_survives_a_weakref_callback_inside_its_own_critical_section():
    """The memo's weak-reference callback takes the same lock as memo(): a garbage collection that frees a keyed tensor
    while memo() holds the lock must not deadlock (it did with a non-reentrant lock: a hang in training)."""
    import gc
    import threading
    import torch
    from pvcnn_amd.modules.functional import _cache

    done = threading.Event()

    def work():
        for i in range(25):
            t = torch.zeros(3)
            _cache.memo(t, 'k', lambda: i)
            cyc = [t]
            cyc.append(cyc)                     # the tensor dies only through the cycle collector ...
            del t, cyc

            def make():
                gc.collect()                    # ... which runs here, inside memo() of another tensor
                return 1
            u = torch.zeros(3)
            with _cache._lock:                  # the callback fires while this thread holds the lock
                gc.collect()
            assert _cache.memo(u, 'k', make) == 1
        done.set()

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(30)
    assert done.is_set(), 'memo() deadlocked against its own weak-reference callback'


# ---- the folded (BatchNorm3d, LeakyReLU, Conv3d) autograd node, wired to a torch stand-in of the native entry points ----------
class _FoldStandIn:
    """The native calls BnActVoxelConv3d makes, evaluated with torch on the CPU: what is under test is the node's WIRING
    (argument order, saved tensors, which gradient goes where), not the kernels (tests/test_gpu_fold.py)."""
    has_bnact_bwd_absmax = True

    def __init__(self):
        self.calls = []

    @staticmethod
    def _act(x, bn):
        import torch
        g, b, mean, rstd, slope = bn
        c = x.shape[1]
        shape = (1, c) + (1,) * (x.dim() - 2)
        scale = (g if g is not None else torch.ones(c)) * rstd
        shift = (b if b is not None else torch.zeros(c)) - mean * scale
        z = x * scale.view(shape) + shift.view(shape)
        return torch.where(z > 0, z, z * slope)

    @staticmethod
    def _bits(t):
        import torch
        return t.abs().max().reshape(1).view(torch.int32)

    def bn_stats(self, x3, rm, rv, momentum, eps):
        import torch
        self.calls.append('bn_stats')
        mean = x3.mean(dim=(0, 2))
        var = x3.var(dim=(0, 2), unbiased=False)
        if rm is not None:
            n = x3.shape[0] * x3.shape[2]
            rm.mul_(1 - momentum).add_(momentum * mean)
            rv.mul_(1 - momentum).add_(momentum * var * n / (n - 1))
        return mean, torch.rsqrt(var + eps)

    def absmax_bits(self, x):
        self.calls.append('absmax_bits')
        return self._bits(x)

    def bnact_absmax_bits(self, x, bn):
        self.calls.append('bnact_absmax_bits')
        return self._bits(self._act(x, bn))

    def conv3d_forward_split_bnact(self, x, weight, bias, bn, want_stats=False, amax=None):
        import torch
        assert amax is not None and torch.equal(amax, self._bits(self._act(x, bn)))
        y = torch.nn.functional.conv3d(self._act(x, bn), weight, bias, padding=1)
        if want_stats:
            return y, torch.zeros(weight.shape[0], 1, 2)
        return y

    def conv3d_backward_data_split(self, grad_y, weight, nsplit, amax=None):
        import torch
        assert nsplit == 2 and torch.equal(amax, self._bits(grad_y))
        b, _, r = grad_y.shape[:3]
        return torch.nn.grad.conv3d_input((b, weight.shape[1], r, r, r), weight, grad_y, padding=1)

    def conv3d_backward_weight_f16_bnact(self, x, grad_y, x_amax, gy_amax, bn, with_bias=False):
        import torch
        assert torch.equal(x_amax, self._bits(self._act(x, bn))) and torch.equal(gy_amax, self._bits(grad_y))
        co, ci = grad_y.shape[1], x.shape[1]
        gw = torch.nn.grad.conv3d_weight(self._act(x, bn), (co, ci, 3, 3, 3), grad_y, padding=1)
        return (gw, grad_y.sum(dim=(0, 2, 3, 4))) if with_bias else gw

    def bnact_backward(self, x3, g3, w, b, mean, rstd, slope, training, want_amax=False):
        import torch
        assert training                                       # batch statistics are differentiated through
        with torch.enable_grad():                             # (called from inside a backward pass)
            xr = x3.detach().clone().requires_grad_()
            wr = w.detach().clone().requires_grad_()
            br = b.detach().clone().requires_grad_()
            m = xr.mean(dim=(0, 2), keepdim=True)
            v = xr.var(dim=(0, 2), unbiased=False, keepdim=True)
            eps = (1.0 / rstd.view(1, -1, 1) ** 2 - v.detach())  # the eps the statistics were finalised with
            z = (xr - m) * torch.rsqrt(v + eps) * wr.view(1, -1, 1) + br.view(1, -1, 1)
            torch.where(z > 0, z, z * slope).backward(g3)
        out = (xr.grad, wr.grad, br.grad)
        return out + (self._bits(xr.grad),) if want_amax else out


def test_folded_batchnorm_conv3d_node_routes_every_gradient(monkeypatch):
    """BnActVoxelConv3d == Conv3d(LeakyReLU(BatchNorm3d(x))) for the output, the input gradient, all four parameter gradients
    and the running statistics; the input gradient leaves the node tagged with its max |.| and the convolution in front of it
    picks that up instead of measuring the tensor again."""
    import torch
    import torch.nn as nn
    from pvcnn_amd.modules.functional import _cache, backend as seam
    from pvcnn_amd.modules.functional.bnact import batch_norm_act_conv3d
    torch.manual_seed(11)
    fake = _FoldStandIn()
    monkeypatch.setattr(seam, '_backend', fake)
    bn, act, conv = nn.BatchNorm3d(6, eps=1e-4), nn.LeakyReLU(0.1), nn.Conv3d(6, 5, 3, padding=1)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    bn2 = nn.BatchNorm3d(6, eps=1e-4)
    bn2.load_state_dict(bn.state_dict())
    x = torch.randn(3, 6, 4, 4, 4)
    x1 = x.clone().requires_grad_()
    y1 = conv(act(bn(x1)))
    w = torch.randn_like(y1)
    (y1 * w).sum().backward()
    want = [x1.grad, bn.weight.grad, bn.bias.grad, conv.weight.grad, conv.bias.grad]
    for p in (bn.weight, bn.bias, conv.weight, conv.bias):
        p.grad = None
    x2 = x.clone().requires_grad_()
    seen = {}

    class Probe(torch.autograd.Function):                    # stands where the first convolution's backward would
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            seen['amax'] = _cache.absmax_of(g, lambda: None)
            seen['bits'] = _FoldStandIn._bits(g)
            return g

    y2, part = batch_norm_act_conv3d(Probe.apply(x2), bn2, 0.1, conv.weight, conv.bias, stats_part=None, want_stats=True)
    assert part.shape[0] == 5 and not part.requires_grad
    assert torch.allclose(y2, y1, atol=1e-5)
    (y2 * w).sum().backward()
    got = [x2.grad, bn2.weight.grad, bn2.bias.grad, conv.weight.grad, conv.bias.grad]
    for a, b in zip(got, want):
        assert a is not None and torch.allclose(a, b, rtol=1e-4, atol=1e-5), (a - b).abs().max()
    assert torch.allclose(bn.running_mean, bn2.running_mean, atol=1e-6) and torch.allclose(bn.running_var, bn2.running_var, atol=1e-6)
    assert int(bn2.num_batches_tracked) == 1
    # the tag travelled with the tensor object from one autograd node to the next
    assert seen['amax'] is not None and torch.equal(seen['amax'], seen['bits'])
    assert fake.calls.count('bnact_absmax_bits') == 1 and fake.calls.count('absmax_bits') == 1   # only y's incoming gradient was measured


def test_absmax_tag_dies_with_an_in_place_update():
    import torch
    from pvcnn_amd.modules.functional import _cache
    t = torch.randn(8)
    tag = torch.tensor([7], dtype=torch.int32)
    _cache.tag_absmax(t, tag)
    assert _cache.absmax_of(t, lambda: None) is tag
    assert _cache.absmax_of(t.view(8), lambda: None) is None          # another tensor object: not tagged
    t.mul_(2.0)                                                       # modified in place: the tag no longer describes it
    assert _cache.absmax_of(t, lambda: None) is None


def test_trace_steady_delimits_steps_by_the_optimizer_not_by_the_gradient_packing(tmp_path):
    """tools/trace_steady.py: the fused-Adam launches end a step; the multi_tensor_apply COPY kernels of the gradient packing
    (pvcnn_amd/dp.py) must not -- counting them as optimizer phases halved every per-step figure once."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    adam = ('void at::native::(anonymous namespace)::multi_tensor_apply_kernel<at::native::(anonymous namespace)::'
            'FusedOptimizerTensorListMetadata<4>, at::native::(anonymous namespace)::FusedAdamMathFunctor<float, 4>, float*>(...)')
    pack = ('void at::native::(anonymous namespace)::multi_tensor_apply_kernel<at::native::(anonymous namespace)::'
            'TensorListMetadata<2>, at::native::(anonymous namespace)::CopyFunctor<float, float, 2, 1, 1>>(...)')
    rows, t = ['Kernel_Name,Start_Timestamp,End_Timestamp'], 0
    for step in range(6):
        for name, dur in (('pvcnn::conv', 300), (pack, 10), ('pvcnn::bn', 50), (pack, 10), (adam, 70), (adam, 70)):
            rows.append(f'"{name}",{t},{t + dur * 1000}')
            t += dur * 1000 + 1000
    trace = tmp_path / 'kernel_trace.csv'
    trace.write_text('\n'.join(rows) + '\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_steady.py'), str(trace), '4', '10', '1'],
                         capture_output=True, text=True, check=True).stdout
    first = out.splitlines()[0]
    assert 'last 4 steps: 24 kernel launches (6/step)' in first, first
    assert 'kernel time 0.510 ms/step' in first, first


def test_bench_byte_formulas_reproduce_the_survey_totals():
    """bench.py's algorithmic bytes (what roofline.achieved is computed from) are SURVEY 8(d)'s formulas: devoxelize forward of the
    R = 32 stage 156.0 MB, backward 155.2 MB, and the eight voxelize + devoxelize calls of a cfg2 step forward + backward 844 MB."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b, n, s32, s16 = 16, 4096, 32 ** 3, 16 ** 3
    assert bench.bytes_devox_fwd(b, 64, n, s32) == 155_975_680
    assert bench.bytes_devox_bwd(b, 64, n, s32) == 155_189_248
    assert bench.bytes_devox_fwd(b, 64, n, s32, training=False) == 155_975_680 - 64 * b * n
    total = bench.bytes_vox_fwd(b, 9, n, s32) + bench.bytes_vox_bwd(b, 9, n, s32)
    total += sum(bench.bytes_vox_fwd(b, c, n, s16) + bench.bytes_vox_bwd(b, c, n, s16) for c in (64, 64, 64))
    total += bench.bytes_devox_fwd(b, 64, n, s32) + bench.bytes_devox_bwd(b, 64, n, s32)
    total += sum(bench.bytes_devox_fwd(b, c, n, s16) + bench.bytes_devox_bwd(b, c, n, s16) for c in (64, 64, 128))
    assert round(total / 1e6) == 844
    # the traffic table: the committed PMC passes, doubled FETCH_SIZE (gfx950), per launch
    t = bench.pmc_traffic('trilinear_devoxelize_fwd', [16, 64, 4096, 32])
    assert t['traffic'] is not None and 1.0 <= t['traffic'] / (bench.bytes_devox_fwd(b, 64, n, s32) + 4 * b * 64 * n) <= 1.1
    assert bench.pmc_traffic('no_such_op', [1, 2, 3, 4]) == {'traffic': None}
