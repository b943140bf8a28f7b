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
