"""gather / furthest_point_sample / logits_mask (reference: modules/functional/sampling.py:8-84)."""
import numpy as np
import torch
from torch.autograd import Function

from ._autograd import native, amp_fwd, amp_bwd

__all__ = ['gather', 'furthest_point_sample', 'logits_mask']


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


def logits_mask(coords, logits, num_points_per_object):
    """Foreground sampling of the Frustum pipeline (reference: sampling.py:51-84).

    coords (B,3,N), logits (B,2,N) -> (selected_coords (B,3,M) centred on the foreground mean,
    foreground mean (B,3), mask (B,N) bool).  Sampling uses numpy's global RNG on the host, as
    the reference does (np.random.choice / shuffle), so seeding numpy reproduces its draws.
    """
    nb, _, npts = coords.shape
    m = int(num_points_per_object)
    mask = logits[:, 0, :] < logits[:, 1, :]
    n_fg = mask.sum(dim=-1, keepdim=True)
    fg_coords = coords * mask.view(nb, 1, npts)
    fg_mean = fg_coords.sum(dim=-1) / torch.max(n_fg, torch.ones_like(n_fg)).float()
    picks = torch.zeros((nb, m), device=coords.device, dtype=torch.int32)
    for bi in range(nb):
        cand = mask[bi].nonzero().view(-1)
        k = cand.numel()
        if k >= m:
            sel = np.random.choice(k, m, replace=False)
        elif k > 0:
            sel = np.concatenate([np.arange(k).repeat(m // k), np.random.choice(k, m % k, replace=False)])
            np.random.shuffle(sel)
        else:
            continue
        picks[bi] = cand[sel]
    return gather(fg_coords - fg_mean.view(nb, -1, 1), picks), fg_mean, mask
