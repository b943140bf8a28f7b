"""concat_points (csrc/bnact.hip): torch.cat along the channels + the amax buffer of the result in one kernel; autograd slices."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('b,n', [(3, 1024), (2, 1000), (2, 777), (1, 4)])
def test_concat_points_equals_torch_cat_and_emits_the_amax_buffer(hip, b, n):
    g = torch.Generator().manual_seed(2)
    wide = torch.randn(b, 40, n, generator=g).to(DEV)
    srcs = [torch.randn(b, 9, n, generator=g).to(DEV), wide[:, 4:20, :], torch.randn(b, 130, n, generator=g).to(DEV),
            torch.randn(b, 7, generator=g).to(DEV).unsqueeze(-1).expand(-1, -1, n)]
    srcs[2][0, 5, n // 2] = -321.0
    out, amax = hip.concat_points(srcs)
    want = torch.cat(srcs, dim=1)
    assert torch.equal(out, want)
    assert torch.equal(amax, hip.absmax_tiles(want.contiguous(), 256))
    out2, none = hip.concat_points(srcs, want_amax=False)
    assert none is None and torch.equal(out2, want)


def test_concat_points_autograd_and_tag(hip):
    from pvcnn_amd import workload
    from pvcnn_amd.modules.functional import _cache
    g = torch.Generator().manual_seed(3)
    b, n = 2, 512
    a = torch.randn(b, 8, n, generator=g).to(DEV).requires_grad_()
    c = torch.randn(b, 5, generator=g).to(DEV).requires_grad_()
    a2, c2 = a.detach().clone().requires_grad_(), c.detach().clone().requires_grad_()
    w = torch.randn(b, 13, n, generator=g).to(DEV)
    out = workload.concat_points([a, c.unsqueeze(-1).expand(-1, -1, n)])
    (out * w).sum().backward()
    ref = torch.cat([a2, c2.unsqueeze(-1).expand(-1, -1, n)], dim=1)
    (ref * w).sum().backward()
    assert torch.equal(out, ref) and torch.equal(a.grad, a2.grad) and torch.allclose(c.grad, c2.grad, rtol=1e-5, atol=1e-5)
    tag = _cache.amax_of(out, 256)
    # (ABI v12: the workload's concatenation asks for the TABLE only -- its consumers, the classifier's GEMMs, are handed the segment
    # length and never read word [0]: tests/test_gpu_amax_table_only.py)
    assert tag is not None and torch.equal(tag[1:], hip.absmax_tiles(ref.detach().contiguous(), 256)[1:])


@pytest.mark.parametrize('b', [8, 1])
def test_the_last_stage_is_written_into_its_slice_of_the_concatenation(hip, b):
    """Round 5: the BatchNorm + ReLU pass of the last point stage writes its output INTO its channel slice of the classifier's
    concatenation (bnact_apply_rowmax(..., out=)), and concat_points copies nothing for that source (in_place=): the same bytes, the
    same amax buffer, the same row maxima as the separate tensor + full copy.  b = 1 (the last batch of an epoch with
    dataset_size % batch_size == 1): the in-place source's cloud stride is the BUFFER's, whatever torch reports for a size-1 dimension."""
    import torch
    g = torch.Generator(device=DEV).manual_seed(11)
    n, c = 1024, 96
    x = torch.randn(b, c, n, device=DEV, generator=g)
    gamma, beta = torch.rand(c, device=DEV, generator=g) + 0.5, torch.randn(c, device=DEV, generator=g)
    mean, rstd = x.mean(dim=(0, 2)), 1.0 / torch.sqrt(x.var(dim=(0, 2), unbiased=False) + 1e-5)
    small = [torch.randn(b, 16, n, device=DEV, generator=g), torch.randn(b, 40, n, device=DEV, generator=g)]
    tail = torch.randn(b, 24, device=DEV, generator=g).unsqueeze(-1).expand(-1, -1, n)
    seg = hip.PW_AMAX_SEG

    def run(in_place):
        whole, armed, keys = hip.amax_and_row_keys(b, c, n, seg, x.device)
        whole.zero_()
        total = 16 + 40 + c + 24
        if in_place:
            buf = torch.full((b, total, n), float('nan'), device=DEV)
            y, winners, values = hip.bnact_apply_rowmax(x, gamma, beta, mean, rstd, 0.0, seg, armed, keys, out=buf[:, 56:56 + c, :])
            assert y.data_ptr() == buf[:, 56:56 + c, :].data_ptr()
            out, amax = hip.concat_points(small + [y, tail], out=buf, in_place={2: armed})
            assert out.data_ptr() == buf.data_ptr()
        else:
            y, winners, values = hip.bnact_apply_rowmax(x, gamma, beta, mean, rstd, 0.0, seg, armed, keys)
            out, amax = hip.concat_points(small + [y, tail])
        return y.clone(), winners, values, out, amax

    ya, wa, va, oa, aa = run(True)
    yb, wb, vb, ob, ab = run(False)
    assert torch.equal(ya, yb) and torch.equal(wa, wb) and torch.equal(va, vb)
    assert torch.equal(oa, ob) and torch.equal(aa, ab)
    assert torch.equal(oa, torch.cat(small + [yb, tail], dim=1)) and torch.equal(wa, yb.max(dim=-1).indices)


def test_pvcnn_with_and_without_the_concatenation_slot_is_the_same_network(hip, monkeypatch):
    """workload.PVCNN / PVCNNShapeNet: the slot path (last stage written in place) against the same model with the slot switched off:
    bit-equal logits and gradients; and the slot path really ran (the concatenation's storage is the buffer the stage wrote)."""
    import copy
    import torch
    from pvcnn_amd import workload
    for build, batch in ((lambda: workload.PVCNN(13, 6, width_multiplier=0.5), lambda: workload.make_s3dis_batch(4, 2048, device=DEV)),
                         (lambda: workload.PVCNNShapeNet(50, 16, 3, width_multiplier=0.25), lambda: workload.make_shapenet_batch(4, 1024, device=DEV)),
                         # one cloud (no BatchNorm1d on (B, C) in this network, so B = 1 trains)
                         (lambda: workload.PVCNNShapeNet(50, 16, 3, width_multiplier=0.25), lambda: workload.make_shapenet_batch(1, 1024, device=DEV))):
        torch.manual_seed(9)
        net = build().to(DEV).train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        twin = copy.deepcopy(net)
        x, y = batch()
        taken = []
        orig = workload.concat_slot

        def spy(*a, **k):
            s = orig(*a, **k)
            taken.append(s)
            return s
        monkeypatch.setattr(workload, 'concat_slot', spy)
        la = torch.nn.functional.cross_entropy(net(x), y)
        la.backward()
        assert taken and taken[-1] is not None
        monkeypatch.setattr(workload, 'concat_slot', lambda *a, **k: None)
        lb = torch.nn.functional.cross_entropy(twin(x), y)
        lb.backward()
        monkeypatch.setattr(workload, 'concat_slot', orig)
        assert la.item() == lb.item()
        for (n1, p1), (n2, p2) in zip(net.named_parameters(), twin.named_parameters()):
            assert torch.equal(p1.grad, p2.grad), n1
