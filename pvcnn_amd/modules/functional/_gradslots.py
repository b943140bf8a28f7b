"""Where a backward kernel may write a parameter's gradient so that nobody has to copy it afterwards.

`pvcnn_amd.dp.GradBucketReducer` keeps the gradients of a model in a few flat buckets (one all-reduce / one optimizer pass per
bucket); until round 4 every backward kernel wrote its weight gradient into a fresh tensor, autograd handed that tensor over as
`p.grad`, and `_Bucket.pack` gathered ~100 of them into the buckets with a multi-tensor copy per bucket (2 x 28 us of a PVCNN step,
9 x 17 us of a PVCNN++ step).  The reducer now registers every parameter's slot here, and the autograd nodes of this package ask
`claim(weight)` for it right before they launch the kernel: the kernel writes into the bucket, the node returns a FRESH ALIAS of the
slot, autograd (no gradient accumulated yet, alias referenced by nobody else) installs it as `p.grad` without a copy, and `pack`
skips it (`g.data_ptr() == view.data_ptr()`).

A slot is handed out at most once per step (`reset()` from `reducer.zero_grad()`), only while `p.grad is None` (otherwise autograd
ADDS what the node returns to the existing gradient: the node must return a tensor of its own) and only outside `create_graph`.  A
parameter used twice in one graph gets the slot for its first use; the second falls back to a fresh tensor and autograd sums the two
as before.  Everything else (torch's own layers, CPU runs, a model without a reducer) never sees this module.
"""
import os
import weakref

import torch

enabled = os.environ.get('PVCNN_GRAD_SLOTS', '1') != '0'      # (debug / A-B switch: 0 = every gradient in a tensor of its own, gathered by pack)

_by_ptr = {}          # data_ptr of a registered parameter -> [weakref(param), its view of the gradient bucket, claimed this step]


def register(param, view):
    """`view`: the slice of a flat gradient bucket shaped like `param` (contiguous).  Re-register after the parameter moved."""
    ent = [weakref.ref(param), view, False]
    _by_ptr[param.data_ptr()] = ent
    return ent


def unregister(entries):
    for ent in entries:
        for k in [k for k, v in _by_ptr.items() if v is ent]:
            del _by_ptr[k]


def reset(entries):
    for ent in entries:
        ent[2] = False


def claim(t):
    """t: a parameter, or a contiguous view of a whole parameter (Conv1d's (Co, Ci, 1) weight seen as (Co, Ci)), whose gradient a
    kernel is about to write -> a new tensor aliasing the parameter's slot in its gradient bucket, shaped like t; or None."""
    if t is None or not _by_ptr or not enabled:
        return None
    ent = _by_ptr.get(t.data_ptr())
    if ent is None or ent[2] or torch.is_grad_enabled():
        return None
    p = ent[0]()
    if (p is None or p.grad is not None or p.data_ptr() != t.data_ptr() or p.numel() != t.numel() or not t.is_contiguous()
            or ent[1].dtype != t.dtype or ent[1].device != t.device):
        return None
    ent[2] = True
    return ent[1].view(t.shape)


def destinations(backend, weight, bias=None):
    """-> the keyword arguments `out_w` / `out_b` for a backend's backward-weight call: the slots of `weight` / `bias` (each may be
    None) where they can be had, nothing for a backend that cannot write into given tensors."""
    if not _by_ptr or not getattr(backend, 'has_grad_out', False):
        return {}
    kw = {}
    w = claim(weight)
    if w is not None:
        kw['out_w'] = w
    b = claim(bias)
    if b is not None:
        kw['out_b'] = b
    return kw
