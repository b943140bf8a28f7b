"""Data-parallel training step for the PVConv path: one process per GPU, RCCL all-reduce over xGMI.

Reference: a single process drives all GPUs through nn.DataParallel (train.py:180-181): per
iteration it scatters inputs, re-broadcasts every parameter, gathers outputs and reduce-adds
gradients to GPU 0.  The path itself shards cleanly over the batch (every kernel indexes
blockIdx.x = cloud; BatchNorm statistics are per replica there too), so here each rank owns one
micro-batch and the ONLY exchange per step is a sum all-reduce of the fp32 gradients, divided by
the world size (= the gradient of the global-batch mean loss for equal shards).

Design for xGMI (point-to-point links, ring collectives are per-link bound):
  * gradients live in a few large FLAT buckets, so a PVCNN step issues 1-3 all-reduces of
    several MiB instead of ~70 small ones (9.8 MiB of gradients in total);
  * buckets are filled in reverse parameter order (the order backward produces gradients): when
    the last gradient of a bucket has arrived, ONE multi-tensor copy packs the bucket's gradients
    into the flat buffer (`p.grad` become views of it) and the all-reduce is launched
    asynchronously from that hook, overlapping RCCL with the rest of backward.  (Letting autograd
    accumulate straight into bucket views instead costs one `add` launch per parameter and a
    memset per bucket: 73 launches, 0.35 ms of a 10 ms PVCNN step on MI355X.)
  * parameters and buffers are broadcast once from rank 0; BN running statistics are not
    synchronised afterwards (DataParallel keeps replica 0's; SyncBN would change the results).

Works with any torch.distributed backend: `nccl` (= RCCL) on GPUs, `gloo` in the CPU tests.
"""
import torch
import torch.distributed as dist

__all__ = ['GradBucketReducer', 'shard_batch']


def shard_batch(global_batch, world_size, rank):
    """Contiguous slice of the global batch owned by `rank` (equal shards required: the mean of
    per-shard mean gradients equals the global mean only then)."""
    if global_batch % world_size:
        raise ValueError(f'global batch {global_batch} is not divisible by world size {world_size}')
    per = global_batch // world_size
    return slice(rank * per, (rank + 1) * per)


class _Bucket:
    __slots__ = ('flat', 'params', 'views', 'pending', 'work', 'packed', 'pflat')

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pflat = None                              # the parameters themselves in the same flat layout (flatten_parameters)
        self.views, off = [], 0
        for p in params:
            self.views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending, self.work, self.packed = len(params), None, False

    def pack(self):
        """Gather this step's gradients into the flat buffer with one multi-tensor copy and make `p.grad` views of it.
        A parameter that received no gradient contributes zeros; one whose gradient already IS its view (accumulated in
        place by a later backward, or zeroed in place by optimizer.zero_grad(set_to_none=False)) is left alone."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                src.append(g.detach())
                dst.append(v)
        if src:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.packed = True


class GradBucketReducer:
    """Bucketed, backward-overlapped gradient all-reduce for one model replica.

    usage per step:
        reducer.zero_grad(); loss = f(model(x)); loss.backward(); reducer.finish(); optimizer.step()
    """

    def __init__(self, model, bucket_mb=8.0, group=None, broadcast=True, always_reduce=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_reduce: issue the collectives even in a 1-rank group (exercises the RCCL path on one GPU)
        self.collective = dist.is_initialized() and (self.world > 1 or always_reduce)
        self._accumulate = False
        # hook-launched (backward-overlapped) all-reduces are issued in the order the buckets fill; that order is the same
        # on every rank as long as every rank uses the same parameters in backward (true for the static graphs of this
        # path).  Set launch_from_hooks = False for models with data-dependent unused parameters: finish() then issues
        # every collective in fixed bucket order (no overlap, no ordering hazard).
        self.launch_from_hooks = True
        self.params = [p for p in model.parameters() if p.requires_grad]
        if self.collective and broadcast:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t, src=0, group=group)
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets, self._owner = [], {}
        chunk, size = [], 0
        for p in reversed(self.params):            # backward produces gradients roughly in this order
            if chunk and (size + p.numel() > cap or p.dtype != chunk[0].dtype or p.device != chunk[0].device):
                self._seal(chunk)
                chunk, size = [], 0
            chunk.append(p)
            size += p.numel()
        if chunk:
            self._seal(chunk)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._slots = []
        self._register_slots()

    def _register_slots(self):
        """Tell this package's backward kernels where each parameter's gradient lives in the buckets (functional/_gradslots.py): they
        write it there, and `_Bucket.pack` has nothing left to copy for those parameters."""
        from .modules.functional import _gradslots
        _gradslots.unregister(self._slots)
        self._slots = [_gradslots.register(p, v) for b in self.buckets for p, v in zip(b.params, b.views)]

    def _seal(self, chunk):
        flat = torch.zeros(sum(p.numel() for p in chunk), dtype=chunk[0].dtype, device=chunk[0].device)
        bucket = _Bucket(flat, chunk)
        for p, v in zip(chunk, bucket.views):
            p.grad = v                                         # defined (zero) gradients from the start, as before
            self._owner[p] = bucket
        self.buckets.append(bucket)

    def _on_grad(self, p):
        b = self._owner[p]
        if self.collective and (b.work is not None or b.pending <= 0):
            # a second backward() before finish(): its local gradients would land on top of an already reduced (or
            # in-flight) bucket and the result would silently be wrong on every rank
            raise RuntimeError('GradBucketReducer: a gradient arrived for a bucket that was already all-reduced in this step; '
                               'call finish() after every backward(), or accumulate under no_sync()')
        if self._accumulate:
            return
        b.pending -= 1
        if b.pending == 0:
            b.pack()
            if self.collective and self.launch_from_hooks:
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def no_sync(self):
        """Context manager for gradient accumulation: backward() passes inside it only accumulate into the buckets; the
        all-reduces are issued by the first backward() outside it (or by finish())."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._accumulate = self._accumulate, True
            try:
                yield
            finally:
                self._accumulate = prev
        return ctx()

    def finish(self):
        """Wait for the in-flight all-reduces, launch those of buckets that never filled (unused
        parameters) and turn sums into means.  Call after backward, before optimizer.step()."""
        for b in self.buckets:
            if not b.packed:                          # never filled (unused parameters), or only accumulated under no_sync()
                b.pack()
        if self.collective:
            # buckets that never filled (parameters unused in this backward): always in fixed bucket order, after the
            # hook-launched ones -- identical on every rank provided the SET of filled buckets is (see launch_from_hooks)
            for b in self.buckets:
                if b.work is None:
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for b in self.buckets:
                b.work.wait()
                if self.world > 1:
                    b.flat.div_(self.world)
        self.rearm()

    def rearm(self):
        """Forget the per-step bucket state (gradients pending, in-flight work, packed flag): the state after finish()."""
        for b in self.buckets:
            b.pending, b.work, b.packed = len(b.params), None, False
        for ent in self._slots:
            ent[2] = False                                     # (a step is over: functional/_gradslots.py may hand every slot out again --
                                                               #  it does so only to a parameter whose `grad` is None by then)

    def zero_grad(self):
        """Forget last step's gradients: `p.grad = None`, so autograd hands every new gradient over as it is (no memset of the
        buckets, no accumulate-add per parameter); `_Bucket.pack` gathers them into the flat buffer when the bucket is complete.
        After finish() `p.grad` are views of the flat buckets again, which is what the (fused) optimizer reads."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None
        for ent in self._slots:
            ent[2] = False                                     # (functional/_gradslots.py: every slot may be handed out once per step)

    def flatten_parameters(self):
        """Lay the PARAMETERS out like their gradients: one flat buffer per bucket, `p.data` becoming views of it (values kept).  An
        optimizer can then update a whole bucket in one elementwise pass (pvcnn_amd.optim.FlatAdam) instead of walking ~100 tensors.
        Call once, after the model sits on its device and before the optimizer / a graph capture is built."""
        with torch.no_grad():
            for b in self.buckets:
                if b.pflat is not None:
                    continue
                b.pflat = torch.empty_like(b.flat)
                off = 0
                for p in b.params:
                    view = b.pflat[off:off + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    off += p.numel()
        self._register_slots()                                 # (keyed by the parameters' addresses, which have just changed)
        return self

    @property
    def gradient_bytes(self):
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)

    def remove(self):
        from .modules.functional import _gradslots
        _gradslots.unregister(self._slots)
        self._slots = []
        for h in self._hooks:
            h.remove()
