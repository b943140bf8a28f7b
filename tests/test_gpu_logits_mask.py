"""logits_mask on the device (reference modules/functional/sampling.py:51-84).

Parity mode: fed numpy's own draws, the device selection is bit-identical to the reference's host loop (all three
outputs, torch.equal).  Device-RNG mode: no host synchronisation, and the selection satisfies the reference's three
cases exactly (distinctness, multiplicities, foreground-only), with a different random stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _inputs(b, n, fg_fraction, seed):
    g = torch.Generator().manual_seed(seed)
    coords = torch.randn(b, 3, n, generator=g)
    logits = torch.randn(b, 2, n, generator=g)
    logits[:, 1] += torch.tensor(fg_fraction).logit() if 0 < fg_fraction < 1 else (50.0 if fg_fraction >= 1 else -50.0)
    return coords.to(DEV), logits.to(DEV)


@pytest.mark.parametrize('b,n,m,frac', [(32, 1024, 512, 0.7), (32, 1024, 512, 0.2), (8, 1000, 512, 0.02), (4, 2048, 128, 0.5),
                                        (3, 777, 100, 1.0), (5, 1024, 512, 0.0), (2, 64, 512, 0.5)])
def test_parity_mode_equals_the_reference_loop(hip, b, n, m, frac):
    from pvcnn_amd.modules.functional.sampling import logits_mask, numpy_choices
    coords, logits = _inputs(b, n, frac, 11)
    if frac not in (0.0, 1.0):
        logits[0, 1] = -50.0                                  # one cloud without any foreground point
    np.random.seed(123)
    want = logits_mask(coords, logits, m, rng='numpy')        # the reference's host loop (checked against the reference
    #                                                           itself on CPU in test_reference_python.py)
    counts = (logits[:, 0] < logits[:, 1]).sum(dim=1).tolist()
    np.random.seed(123)
    choices = torch.from_numpy(numpy_choices(counts, m))
    got = logits_mask(coords, logits, m, choices=choices)
    for a, e in zip(got, want):
        assert torch.equal(a, e)


def test_device_rng_mode_has_no_host_sync_and_the_reference_cases(hip):
    from pvcnn_amd.modules.functional.sampling import logits_mask
    b, n, m = 16, 1024, 512
    coords, logits = _inputs(b, n, 0.6, 5)
    logits[1, 1, :] = -50.0; logits[1, 1, :100] = 50.0        # k = 100 < M: repeats + extras
    logits[2, 1, :] = -50.0                                   # k = 0
    logits[3, 1, :] = -50.0; logits[3, 1, 7] = 50.0           # k = 1
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')                   # any device -> host synchronisation raises
    try:
        sel_coords, mean, mask = logits_mask(coords, logits, m)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    be = hip
    picks = be.mask_select(mask.contiguous(), m, seed=torch.tensor([42, 0], device=DEV))
    picks2 = be.mask_select(mask.contiguous(), m, seed=torch.tensor([42, 0], device=DEV))
    picks3 = be.mask_select(mask.contiguous(), m, seed=torch.tensor([43, 0], device=DEV))
    assert torch.equal(picks, picks2) and not torch.equal(picks, picks3)      # a function of the seed
    mask_c, picks_c = mask.cpu(), picks.cpu().long()
    for i in range(b):
        k = int(mask_c[i].sum())
        row = picks_c[i]
        if k == 0:
            assert (row == 0).all()
            continue
        assert mask_c[i][row].all(), 'only foreground points may be selected'
        counts = torch.bincount(row, minlength=n)[mask_c[i]]
        if k >= m:
            assert counts.max() == 1 and counts.sum() == m                    # M distinct points
        else:
            assert counts.min() >= m // k and counts.max() <= m // k + 1 and counts.sum() == m
            assert (counts == m // k + 1).sum() == m % k                       # exactly M % k extra distinct ones
    # selection is not just "the first M": positions are spread over the candidate list and in random order
    row = picks_c[0]
    assert not torch.equal(row, row.sort().values)
    assert sel_coords.shape == (b, 3, m) and mean.shape == (b, 3)
    assert torch.isfinite(sel_coords).all()
