"""ABI v11: the finalize step of three reductions CAN be the tail of the reduction's own launch ("last workgroup done" tickets,
csrc/common.h) instead of a ~5 us launch behind it: the per-channel sums of the BatchNorm backward, the global maximum of an amax
buffer behind pvcnn_absmax_tiles and behind pvcnn_concat_points.  Bit-identical to the separate launches (NULL tickets through the
C ABI: the default since the A/B of round 5 -- the atomics of ~100 k workgroups per step cost what the launches cost), on repeated
calls that reuse the ticket words, and the tickets are left zeroed."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _both(hip, fn):
    """-> (fn() with the finalize step folded into the reduction's launch, fn() with the separate launches: the default -- the fold
    was measured and lost, profiles/ab/r05d_fold_without_fences.md; it stays selectable and pinned here)."""
    hip.fold_finalize = True
    try:
        folded = fn()
    finally:
        hip.fold_finalize = False
    try:
        separate = fn()
    finally:
        del hip.fold_finalize
    torch.cuda.synchronize()
    return folded, separate


def _tickets_clean(hip):
    pools = hip.__dict__.get('_ticket_pools', {})
    assert pools, 'no ticket pool was created: the folded path did not run'
    for pool, _ in pools.values():
        assert int(pool.abs().max()) == 0


@pytest.mark.parametrize('b,c,s,seg', [(16, 64, 32768, 32), (16, 1024, 4096, 256), (16, 64, 4096, 16), (3, 7, 1000, 0), (2, 130, 8200, 0), (8, 512, 2048, 256)])
def test_batchnorm_backward_finalises_its_own_sums(hip, b, c, s, seg):
    g = torch.Generator(device=DEV).manual_seed(b * 1000 + c)
    x = torch.randn(b, c, s, device=DEV, generator=g) * 3 + 0.5
    gy = torch.randn(b, c, s, device=DEV, generator=g)
    gamma = torch.randn(c, device=DEV, generator=g)
    beta = torch.randn(c, device=DEV, generator=g)
    mean = x.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(x.var(dim=(0, 2), unbiased=False) + 1e-4)
    for rep in range(3):                                   # the same ticket words serve every call
        fold, sep = _both(hip, lambda: hip.bnact_backward(x, gy, gamma, beta, mean, rstd, 0.1, True, amax_seg=seg))
        assert len(fold) == len(sep) == (4 if seg else 3)
        for a, b_ in zip(fold, sep):
            assert torch.equal(a, b_)
    _tickets_clean(hip)
    # ... and a channel-slice view as grad_y (torch.cat's backward) takes the same path
    wide = torch.randn(b, 3 * c, s, device=DEV, generator=g)
    view = wide[:, c:2 * c, :]
    fold, sep = _both(hip, lambda: hip.bnact_backward(x, view, gamma, beta, mean, rstd, 0.0, True, amax_seg=seg))
    assert all(torch.equal(a, b_) for a, b_ in zip(fold, sep))
    want = hip.bnact_backward(x, view.contiguous(), gamma, beta, mean, rstd, 0.0, True, amax_seg=seg)
    assert all(torch.equal(a, b_) for a, b_ in zip(fold, want))


def test_folded_sums_are_the_fp64_sums(hip):
    """Not only equal to the old launches: dgamma / dbeta against an fp64 evaluation (<= 2e-6 of the largest)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    b, c, s = 16, 128, 4096
    x = torch.randn(b, c, s, device=DEV, generator=g)
    gy = torch.randn(b, c, s, device=DEV, generator=g)
    gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    mean = x.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(x.var(dim=(0, 2), unbiased=False) + 1e-5)
    hip.fold_finalize = True
    try:
        gx, gg, gb = hip.bnact_backward(x, gy, gamma, beta, mean, rstd, 0.0, True)
    finally:
        del hip.fold_finalize
    xd, gd = x.double(), gy.double()
    xhat = (xd - mean.double().view(1, -1, 1)) * rstd.double().view(1, -1, 1)
    gp = gd * (xhat > 0)
    assert (gb.double() - gp.sum(dim=(0, 2))).abs().max() <= 2e-6 * gp.sum(dim=(0, 2)).abs().max()
    assert (gg.double() - (gp * xhat).sum(dim=(0, 2))).abs().max() <= 2e-6 * (gp * xhat).sum(dim=(0, 2)).abs().max()


@pytest.mark.parametrize('b,c,l,seg', [(16, 64, 32768, 32), (16, 9, 32768, 32), (16, 1472, 4096, 256), (2, 3, 1001, 16), (1, 1, 5, 4), (8, 64, 4096, 16)])
def test_amax_buffer_global_maximum_from_the_table_pass(hip, b, c, l, seg):
    g = torch.Generator(device=DEV).manual_seed(l + c)
    x = torch.randn(b, c, l, device=DEV, generator=g)
    x[b // 2, c // 2, l // 3] = -77.5                        # the global maximum sits in one segment
    for rep in range(3):
        fold, sep = _both(hip, lambda: hip.absmax_tiles(x, seg))
        assert torch.equal(fold, sep)
    assert fold[0].item() == torch.tensor(77.5).view(torch.int32).item()
    _tickets_clean(hip)


@pytest.mark.parametrize('b,n,chans', [(16, 4096, (64, 64, 64, 128, 1024, 128)), (8, 2048, (16, 64, 128, 128, 512, 2048, 2048)), (2, 300, (3, 5))])
def test_concat_points_global_maximum_from_the_copy(hip, b, n, chans):
    g = torch.Generator(device=DEV).manual_seed(n)
    taps = [torch.randn(b, c, n, device=DEV, generator=g) for c in chans[:-1]]
    taps.append(torch.randn(b, chans[-1], device=DEV, generator=g).unsqueeze(-1).expand(-1, -1, n))    # a broadcast source
    for rep in range(2):
        (out_f, am_f), (out_s, am_s) = _both(hip, lambda: hip.concat_points(taps))
        assert torch.equal(out_f, out_s) and torch.equal(am_f, am_s)
    assert torch.equal(out_f, torch.cat(taps, dim=1))
    assert am_f[0].item() == torch.cat(taps, dim=1).abs().max().view(torch.int32).item()
    _tickets_clean(hip)


def test_fold_switch_reaches_the_c_abi(hip, monkeypatch):
    """Without `fold_finalize` the calls pass NULL tickets (no pool is even created); with it a folded call advances the pool's cursor."""
    x = torch.randn(4, 8, 1024, device=DEV)
    assert not hip.fold_finalize                        # the default: NULL tickets, no pool
    hip.absmax_tiles(x, 256)
    assert not hip.__dict__.get('_ticket_pools')
    hip.fold_finalize = True
    try:
        hip.absmax_tiles(x, 256)
        cursor = next(iter(hip._ticket_pools.values()))[1]               # (one pool per (device, stream) since round 6)
        hip.absmax_tiles(x, 256)
        assert next(iter(hip._ticket_pools.values()))[1] > cursor
    finally:
        del hip.fold_finalize
