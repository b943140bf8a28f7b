"""The weight-gradient launches of backward on a stream of their own.

A backward node of this package computes two independent products from the incoming gradient: the gradient of its INPUT, which the
next node is waiting for, and the gradient of its WEIGHT, which nobody reads before the optimizer (or the gradient all-reduce of its
bucket).  On one stream the second sits on the critical path: split-K partials, their reduce launch, every launch a dependent kernel
node.  Issued on a side stream -- behind everything the main stream has issued so far, so it sees the same `x` / `grad_y` -- it is a
parallel path: eagerly a second queue of the device, under a graph capture a branch of the graph (the side stream joins the capture
through the fork event), which the chip fills into the tails and the latency-bound small launches of the main chain.

Only where nothing on the main stream touches the result before a known join:
  * the node writes the gradient INTO the parameter's slot of a flat gradient bucket (`_gradslots.claim`: a reducer is present, the
    slot is free): autograd installs the returned alias as `p.grad` without a kernel, `_Bucket.pack` skips it;
  * `pvcnn_amd.dp.GradBucketReducer` joins (`join()`: the current stream waits for the side stream) before it packs / all-reduces a
    complete bucket and at the top of `finish()`; `pvcnn_amd.graph.GraphedTrainStep` joins behind `backward()`, inside the capture.
Anything else (no reducer, slot taken, create_graph, CPU) runs in line as before.  `PVCNN_WGRAD_PATH=0` (read once) switches it off.

Memory: tensors of the main stream's pool that the side launches read are `record_stream`ed -- the caching allocator does not hand
their blocks out again before the side stream has passed them (under a capture: not before the capture ends); what the launches
allocate themselves (split-K scratch) comes from the side stream's pool.  Same kernels, same operands, same bits.
"""
import contextlib
import os

import torch

__all__ = ['enabled', 'usable', 'forked', 'join']

enabled = os.environ.get('PVCNN_WGRAD_PATH', '1') != '0'
_side = {}        # device index -> torch.cuda.Stream
_open = set()     # device indices with side work the main stream has not waited for yet


def usable(destinations, want_bias, ref):
    """May this node's backward-weight launch go to the side stream?  `destinations`: what `_gradslots.destinations` handed out."""
    return (enabled and ref.is_cuda and 'out_w' in destinations and (not want_bias or 'out_b' in destinations))


@contextlib.contextmanager
def forked(ref, reads=(), on=True):
    """with forked(grad_y, (x, grad_y, ...)): the body's launches go to the side stream of `ref`'s device, ordered behind everything
    issued on the current stream so far.  `reads`: tensors (or None) the body's kernels read.  on=False: a no-op context."""
    if not on:
        yield
        return
    idx = ref.device.index if ref.device.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(idx)
    side = _side.get(idx)
    if side is None:
        side = _side[idx] = torch.cuda.Stream(device=idx)
    side.wait_stream(main)
    _open.add(idx)
    with torch.cuda.stream(side):
        yield
    for t in reads:
        if t is not None and t.is_cuda:
            t.record_stream(side)


def join():
    """The current stream of every device with open side work waits for it."""
    for idx in list(_open):
        torch.cuda.current_stream(idx).wait_stream(_side[idx])
        _open.discard(idx)
