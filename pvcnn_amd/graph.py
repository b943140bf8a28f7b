"""One training step as a hipGraph.

A PVCNN step is ~380 kernel launches of 5-400 us.  Since the dense convolutions moved to the 16-bit matrix cores the GPU finishes
them faster than one Python thread can issue them (ctypes call + torch dispatch: ~25 us per launch), i.e. the eager step is
host-bound.  The step has static shapes and no host synchronisation (the per-coords plans, the device-side logits_mask and the
BatchNorm statistics all stay on the device), so it is captured once -- zero the gradient buckets, forward, loss, backward and,
on one GPU, the fused Adam update -- and replayed with a single launch per step.

Multi-GPU: only forward + backward are captured; the bucket all-reduces (RCCL) and the optimizer run eagerly after the replay
(`GradBucketReducer.finish()` issues them in fixed bucket order).  PVCNN's gradients are a few MB, so nothing is lost by not
overlapping them with backward, and no collective has to live inside a captured graph.

New input data goes INTO the static tensors the step was captured with (`tensor.copy_(batch)`), as with any captured graph.
"""
import contextlib

import torch

from .modules.functional import _cache

__all__ = ['GraphedTrainStep']


class GraphedTrainStep:
    """step = GraphedTrainStep(model, loss_fn, optimizer, reducer); loss = step()

    loss_fn() -> scalar loss tensor, reading the model's inputs / targets from static device tensors.
    optimizer: built with capturable=True when it is to be captured (single GPU).
    capture=False (the default without a GPU): the same sequencing -- forward + backward with the reducer's collectives held back,
    then finish() in fixed bucket order and the optimizer -- issued eagerly; this is what the world-size-2 gloo test drives.
    """

    def __init__(self, model, loss_fn, optimizer, reducer, autocast=contextlib.nullcontext, warmup=3, capture=None):
        self.model, self.loss_fn, self.optimizer, self.reducer, self.autocast = model, loss_fn, optimizer, reducer, autocast
        self.collective = bool(reducer.collective)
        self.capture = torch.cuda.is_available() if capture is None else bool(capture)
        self.graph = self.loss = None
        if self.collective:
            reducer.launch_from_hooks = False            # collectives are issued by finish(), outside the graph
        elif self.capture and not all(g.get('capturable', False) for g in optimizer.param_groups):
            # a non-capturable optimizer keeps its step counter on the host: the captured update would replay step 1 forever
            raise ValueError('GraphedTrainStep captures optimizer.step(): build the optimizer with capturable=True')
        # the f16x2 weight images of every layer from one launch per kind and step (backend.weight_bank_refresh) instead of one per layer
        from .modules.functional._autograd import native
        be = native() if torch.cuda.is_available() else None
        self.bank = be if (be is not None and getattr(be, 'has_weight_bank', False) and next(model.parameters()).is_cuda) else None
        if self.bank is not None:
            self.bank.weight_bank_register(model)
        if not self.capture:
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm up on the capture stream's side: lazy inits, allocator pools
            for _ in range(max(1, warmup)):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _cache.clear()                                   # the per-coords plans must be rebuilt INSIDE the graph
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._forward_backward()
            if not self.collective:
                self.reducer.finish()
                self.optimizer.step()
        if self.collective:
            # the capture ran the reducer's hooks (pending -> 0, packed) but not finish(): re-arm the buckets so that an eager
            # backward after construction does not trip the "already all-reduced" guard
            self.reducer.rearm()
        _cache.clear()                                   # nothing outside may alias tensors of the graph's private pool

    def _forward_backward(self):
        self.reducer.zero_grad()
        if self.bank is not None:
            self.bank.weight_bank_refresh()
        with self.autocast():
            loss = self.loss_fn()
        loss.backward()
        return loss

    def eager_step(self):
        loss = self._forward_backward()
        self.reducer.finish()
        self.optimizer.step()
        return loss

    def __call__(self):
        if self.graph is None:
            return self.eager_step()
        self.graph.replay()
        if self.collective:
            self.reducer.finish()
            self.optimizer.step()
        return self.loss
