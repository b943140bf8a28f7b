"""gather / furthest_point_sample / logits_mask (reference: modules/functional/sampling.py:8-84)."""
import numpy as np
import torch
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['gather', 'furthest_point_sample', 'logits_mask', 'numpy_choices']


class Gather(Function):
    """features (B,C,N), indices (B,M) -> (B,C,M); backward scatter-adds into (B,C,N)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, features, indices):
        feats = features.contiguous()
        idx = indices.int().contiguous()
        ctx.save_for_backward(idx)
        ctx.num_points = feats.size(-1)
        return native().gather_features_forward(feats, idx)

    @staticmethod
    @amp_bwd
    def backward(ctx, grad_gathered):
        idx, = ctx.saved_tensors
        return native().gather_features_backward(grad_gathered.contiguous(), idx, ctx.num_points), None


gather = Gather.apply


def furthest_point_sample(coords, num_samples):
    """coords (B,3,N) -> coordinates (B,3,M) of M iteratively-furthest points (starting at point 0)."""
    coords = coords.contiguous()
    picked = native().furthest_point_sampling(coords, num_samples)
    return gather(coords, picked)


LOGITS_MASK_RNG = 'device'     # GPU default; 'numpy' = the reference's host loop (numpy global RNG, one sync per cloud)


def numpy_choices(counts, num_points_per_object):
    """The reference's numpy draws (sampling.py:74-82) for foreground counts `counts` (iterable of B ints), as a (B, M)
    int32 array of positions into each cloud's ascending foreground list -- consumes np.random exactly like the
    reference's loop does, so under the same np.random.seed it reproduces its selection.  Rows with count 0 are 0."""
    m = int(num_points_per_object)
    rows = []
    for k in counts:
        k = int(k)
        if k >= m:
            sel = np.random.choice(k, m, replace=False)
        elif k > 0:
            sel = np.concatenate([np.arange(k).repeat(m // k), np.random.choice(k, m % k, replace=False)])
            np.random.shuffle(sel)
        else:
            sel = np.zeros(m, dtype=np.int64)
        rows.append(sel)
    return np.stack(rows).astype(np.int32) if rows else np.zeros((0, m), dtype=np.int32)


def logits_mask(coords, logits, num_points_per_object, rng=None, choices=None):
    """Foreground sampling of the Frustum pipeline (reference: sampling.py:51-84).

    coords (B,3,N), logits (B,2,N) -> (selected_coords (B,3,M) centred on the foreground mean,
    foreground mean (B,3), mask (B,N) bool).

    The mask, the foreground mean and the final gather are the reference's torch ops.  The index selection in between --
    in the reference a Python loop with a `nonzero()` sync and numpy draws per cloud -- runs
      * on the GPU (default, `rng='device'`): one kernel per batch (csrc/mask_select.hip), Philox stream seeded from
        torch's device generator, NO host synchronisation; same three cases and distribution as the reference, different
        random numbers than numpy's;
      * with `choices` (B,M) int32 (positions into each cloud's foreground list, e.g. `numpy_choices(counts, M)`): the
        same kernel in parity mode -- bit-identical to the reference for the same draws;
      * with `rng='numpy'`, and always for CPU tensors: the reference's host loop (numpy's global RNG), so seeding numpy
        reproduces its draws.
    """
    nb, _, npts = coords.shape
    m = int(num_points_per_object)
    mask = logits[:, 0, :] < logits[:, 1, :]
    n_fg = mask.sum(dim=-1, keepdim=True)
    fg_coords = coords * mask.view(nb, 1, npts)
    fg_mean = fg_coords.sum(dim=-1) / torch.max(n_fg, torch.ones_like(n_fg)).float()
    be = native()
    mode = rng or LOGITS_MASK_RNG
    on_device = coords.is_cuda and getattr(be, 'has_mask_select', False) and max(npts, m) <= 8192
    if on_device and choices is not None:
        picks = be.mask_select(mask.contiguous(), m, choices=choices.to(device=coords.device, dtype=torch.int32).contiguous())
    elif on_device and mode == 'device':
        seed = torch.randint(0, 2 ** 62, (2,), device=coords.device, dtype=torch.int64)     # device generator: no sync
        picks = be.mask_select(mask.contiguous(), m, seed=seed)
    else:
        picks = torch.zeros((nb, m), device=coords.device, dtype=torch.int32)
        for bi in range(nb):
            cand = mask[bi].nonzero().view(-1)
            k = cand.numel()
            if k == 0:
                continue
            sel = choices[bi].cpu().numpy() if choices is not None else numpy_choices([k], m)[0]
            picks[bi] = cand[torch.as_tensor(sel, dtype=torch.long, device=cand.device)]
    return gather(fg_coords - fg_mean.view(nb, -1, 1), picks), fg_mean, mask
