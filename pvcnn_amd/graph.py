"""One training step as a hipGraph.

A PVCNN step is ~380 kernel launches of 5-400 us.  Since the dense convolutions moved to the 16-bit matrix cores the GPU finishes
them faster than one Python thread can issue them (ctypes call + torch dispatch: ~25 us per launch), i.e. the eager step is
host-bound.  The step has static shapes and no host synchronisation (the per-coords plans, the device-side logits_mask and the
BatchNorm statistics all stay on the device), so it is captured once -- zero the gradient buckets, forward, loss, backward and,
on one GPU, the fused Adam update -- and replayed with a single launch per step.

Multi-GPU: the WHOLE step is captured as well, collectives included (`mode == 'graph+collectives'`): the reducer's autograd hooks
launch each bucket's RCCL all-reduce while backward is being captured, so inside the graph the collective of a full bucket runs on
RCCL's stream next to the rest of backward (the overlap of the eager path, `pvcnn_amd/dp.py`), `finish()` contributes the stream
joins and the division by the world size, and the fused Adam update follows -- one graph launch per step on every rank, the same
RCCL kernels in the same order everywhere.  If the collectives cannot be captured (an RCCL build without capture support), the
fallback captures forward + backward only and issues the bucket all-reduces in fixed bucket order and the optimizer after each
replay (`mode == 'graph, collectives after replay'`: not overlapped).

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
    capture=False (the default without a GPU): the same sequencing issued eagerly -- capture_collectives=True: the hooks launch each
    full bucket's all-reduce during backward (the order the captured graph holds them in); otherwise the collectives are held back
    and finish() issues them in fixed bucket order -- then the optimizer; this is what the world-size-2 gloo tests drive.
    """

    def __init__(self, model, loss_fn, optimizer, reducer, autocast=contextlib.nullcontext, warmup=3, capture=None,
                 capture_collectives=None):
        """capture_collectives (multi-rank / always_reduce only): None = try to capture the RCCL all-reduces inside the graph and
        fall back to issuing them after each replay if that fails; True = capture them or raise; False = always after the replay."""
        self.model, self.loss_fn, self.optimizer, self.reducer, self.autocast = model, loss_fn, optimizer, reducer, autocast
        self.collective = bool(reducer.collective)
        self.capture = torch.cuda.is_available() if capture is None else bool(capture)
        self.graph = self.loss = None
        self.whole_step = not self.collective            # does the graph hold finish() + optimizer.step() too?
        self._sync_hyper = getattr(optimizer, 'sync_hyperparameters', None)      # pvcnn_amd.optim.FlatAdam
        self.capture_error = None
        if self.capture and not all(g.get('capturable', False) for g in optimizer.param_groups) and (
                not self.collective or capture_collectives is not False):
            # a non-capturable optimizer keeps its step counter on the host: the captured update would replay step 1 forever
            raise ValueError('GraphedTrainStep captures optimizer.step(): build the optimizer with capturable=True')
        # the f16x2 weight images of every layer from one launch per kind and step (backend.weight_bank_refresh) instead of one per layer
        from .modules.functional._autograd import native
        be = native() if torch.cuda.is_available() else None
        self.bank = be if (be is not None and getattr(be, 'has_weight_bank', False) and next(model.parameters()).is_cuda) else None
        if self.bank is not None:
            self.bank.weight_bank_register(model)
        if not self.capture:
            if self.collective:
                # the same sequencing as the captured modes, issued eagerly (tests/test_dp_gloo.py drives both on gloo): hooks launch
                # each full bucket's all-reduce during backward (capture_collectives=True) or finish() issues them afterwards
                reducer.launch_from_hooks = bool(capture_collectives)
                self.whole_step = bool(capture_collectives)
            return
        self._warm_up(warmup)
        self._capture_agreed(capture_collectives)

    def _warm_up(self, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm up on the capture stream's side: lazy inits, allocator pools, and
            for _ in range(max(1, warmup)):              # (multi-rank) the RCCL communicator, which must exist before a capture
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    @staticmethod
    def _sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _capture_agreed(self, capture_collectives):
        """Capture the step in the best mode EVERY rank can take (tests/test_dp_gloo.py drives this with a capture that fails on one
        rank only)."""
        reducer = self.reducer
        if self.collective and capture_collectives is not False:
            failure = None
            try:
                self._capture(whole_step=True)
            except Exception as exc:                     # noqa: BLE001 -- reported (mode / capture_error); the fallback still trains
                failure = exc
                self.graph = None
                self.reducer.rearm()
                self._sync()
            # THE RANKS DECIDE TOGETHER: a rank that replays a graph with the collectives inside and a rank that issues
            # them after its replay post different RCCL sequences and hang.  The outcome is agreed outside any capture -- a MIN
            # all-reduce of the success flag over the reducer's group -- and EVERY rank falls back if any rank failed (an aborted
            # capture has executed nothing: its all-reduces were recorded, never launched, so the communicator is where the warm-up
            # steps left it).
            if not self._all_ranks(failure is None):
                if capture_collectives:                  # "capture them or raise": raise on every rank, not only on the one that failed
                    raise (failure or RuntimeError('GraphedTrainStep: the capture of the collectives failed on another rank'))
                self.capture_error = (f'{type(failure).__name__}: {str(failure)[:200]}' if failure is not None
                                      else 'the capture of the collectives failed on another rank')
                if self.graph is not None:               # this rank's capture worked: drop it, take the agreed mode
                    self.graph = None
                    self.reducer.rearm()
                    self._sync()
        if self.graph is None:
            if self.collective:
                reducer.launch_from_hooks = False        # collectives are issued by finish(), after the replay, in fixed bucket order
            try:
                self._capture(whole_step=not self.collective)
            except Exception:
                self._all_ranks(False)                   # the others learn it here instead of hanging in their first collective
                raise
        if self.collective and not self._all_ranks(True):
            raise RuntimeError('GraphedTrainStep: another rank failed to capture the step')

    def _all_ranks(self, ok):
        """True iff `ok` on every rank of the reducer's group (one MIN all-reduce, outside any capture; 1 rank / no group: `ok`)."""
        import torch.distributed as dist
        if not (self.collective and dist.is_initialized() and self.reducer.world > 1):
            return bool(ok)
        dev = next(self.model.parameters()).device
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.reducer.group)
        return bool(flag.item())

    def _quiesce(self):
        """Drain everything the process group's watchdog could still be polling before a capture starts: the device, then every
        outstanding collective of the warm-up steps (the reducer's own Work handles are waited on in finish(); a barrier's completion
        on every rank orders this rank behind the others' warm-up collectives too)."""
        import torch.distributed as dist
        torch.cuda.synchronize()
        if dist.is_initialized() and self.reducer.world > 1:
            dist.barrier(group=self.reducer.group)
            torch.cuda.synchronize()

    def _capture(self, whole_step):
        _cache.clear()                                   # the per-coords plans must be rebuilt INSIDE the graph
        graph = torch.cuda.CUDAGraph()
        mode = 'global'
        if self.collective:
            # A process group runs a watchdog thread that polls the events of collectives issued so far (the warm-up steps').  In the
            # default 'global' capture mode ANY thread's hipEventQuery while this thread captures is an error -- seen as a process
            # abort in the middle of a capture, once in ~4 runs (gpurun_out/r04/gpu_tests.log).  'thread_local' restricts the check to
            # the capturing thread (the watchdog's polls have nothing to do with this capture) -- THAT is what makes the capture safe;
            # the device and the group are drained first (_quiesce: deterministic, no sleep) so that the watchdog has nothing new to poll.
            mode = 'thread_local'
            self._quiesce()
        with torch.cuda.graph(graph, capture_error_mode=mode):
            self.loss = self._forward_backward()         # (multi-rank, whole step: the hooks launch the bucket all-reduces in here)
            if whole_step:
                self.reducer.finish()
                self.optimizer.step()
        self.graph, self.whole_step = graph, whole_step
        if not whole_step:
            # the capture ran the reducer's hooks (pending -> 0, packed) but not finish(): re-arm the buckets so that an eager
            # backward after construction does not trip the "already all-reduced" guard
            self.reducer.rearm()
        _cache.clear()                                   # nothing outside may alias tensors of the graph's private pool

    @property
    def mode(self):
        if self.graph is None:
            return 'eager'
        if not self.collective:
            return 'graph'
        return 'graph+collectives' if self.whole_step else 'graph, collectives after replay'

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
        if self._sync_hyper is not None:
            self._sync_hyper()                           # a scheduler moved the learning rate: the captured update reads it from the device
        self.graph.replay()
        if not self.whole_step:
            self.reducer.finish()
            self.optimizer.step()
        return self.loss
