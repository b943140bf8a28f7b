"""Adam on the flat buckets of the data-parallel step (csrc/optim.hip).

Reference: train.py:96-119 calls `optimizer.step()` of torch.optim.Adam on the model's ~100 parameter tensors, train.py:121-122
`scheduler.step()` of a torch.optim.lr_scheduler on that optimizer, and train.py:186-199 / 249-255 save and restore
`optimizer.state_dict()`.  Here the parameters are laid out like the gradient buckets of `pvcnn_amd.dp.GradBucketReducer`
(flatten_parameters) and one elementwise kernel updates a whole bucket: same arithmetic as torch.optim.Adam (no amsgrad; weight decay
added to the gradient), fp32, step counter AND hyper-parameters on the device, so the update is graph-capturable (pvcnn_amd/graph.py)
and costs ~10 us instead of 2 x 72 us for PVCNN's 9.8 MiB.

FlatAdam IS a torch.optim.Optimizer: `param_groups` / `state` / `state_dict()` / `load_state_dict()` have torch.optim.Adam's layout
(per-parameter `step`, `exp_avg`, `exp_avg_sq` -- here views of the flat moment buffers), so a checkpoint written by either loads into
the other, and a `torch.optim.lr_scheduler` drives it unchanged: the scheduler writes `param_groups[0]['lr']` on the host, and the next
`step()` -- or, for a captured step, `GraphedTrainStep.__call__` through `sync_hyperparameters()` -- copies the five hyper-parameters
into the device block the kernel reads.

GPU only (there is no CPU implementation of the kernel); use torch.optim.Adam for CPU runs.
"""
import ctypes

import torch

from . import _lib

__all__ = ['FlatAdam']


class FlatAdam(torch.optim.Optimizer):
    """opt = FlatAdam(reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0);  per step, after reducer.finish(): opt.step().

    The gradients are read from the reducer's flat buckets (where `finish()` leaves them: `p.grad` are views of those buffers).
    One parameter group (the reference trains with one: train.py:163)."""

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not reducer.buckets or not all(b.flat.is_cuda and b.flat.dtype == torch.float32 for b in reducer.buckets):
            raise ValueError('FlatAdam needs float32 parameters on a GPU (use torch.optim.Adam elsewhere)')
        reducer.flatten_parameters()
        self.reducer = reducer
        defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay),
                        capturable=True)
        super().__init__(list(reducer.params), defaults)         # model.parameters() order: torch.optim.Adam(model.parameters())'s
        dev = reducer.buckets[0].flat.device
        self.exp_avg = [torch.zeros_like(b.flat) for b in reducer.buckets]
        self.exp_avg_sq = [torch.zeros_like(b.flat) for b in reducer.buckets]
        self.step_count = torch.zeros((1,), dtype=torch.float32, device=dev)
        self.hyper = torch.zeros((5,), dtype=torch.float32, device=dev)     # lr, beta1, beta2, eps, weight_decay: what the kernel reads
        self._hyper_host = None
        # torch.optim.Adam's per-parameter state, as views: `step` is the one device counter, the moments are slices of the flat buffers
        for i, b in enumerate(reducer.buckets):
            off = 0
            for p in b.params:
                n = p.numel()
                self.state[p] = {'step': self.step_count.view(()), 'exp_avg': self.exp_avg[i][off:off + n].view_as(p),
                                 'exp_avg_sq': self.exp_avg_sq[i][off:off + n].view_as(p)}
                off += n
        self._lib = _lib.load()
        self.sync_hyperparameters()

    def add_param_group(self, param_group):
        if getattr(self, 'reducer', None) is not None and self.param_groups:
            raise NotImplementedError('FlatAdam updates the ONE parameter group its reducer laid out (the reference trains with one: '
                                      'train.py:163); build a second reducer / optimizer pair for other parameters')
        super().add_param_group(param_group)

    def _group_hyper(self):
        g = self.param_groups[0]
        return (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']))

    def sync_hyperparameters(self):
        """Copy param_groups[0]'s lr / betas / eps / weight_decay into the device block when they changed on the host (a scheduler
        stepped, a checkpoint was loaded).  One small host-to-device copy, only then; never call it while a graph is being captured
        (step() skips it there -- the captured kernel reads whatever the block holds at replay time)."""
        now = self._group_hyper()
        if now != self._hyper_host:
            self.hyper.copy_(torch.tensor(now, dtype=torch.float32), non_blocking=False)
            self._hyper_host = now

    def _check_aliasing(self):
        for b in self.reducer.buckets:
            lo = b.pflat.data_ptr()
            hi = lo + b.pflat.numel() * 4
            for p in b.params:
                if not (lo <= p.data_ptr() < hi):
                    raise RuntimeError('FlatAdam: a parameter no longer lives in its flat bucket (model.to() / .half() / .float() after the '
                                       'optimizer was built re-allocates parameters): build the reducer and the optimizer last')

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise RuntimeError('FlatAdam updates one parameter group')
        vp = ctypes.c_void_p
        last = len(self.reducer.buckets) - 1
        dev = self.step_count.device
        with torch.cuda.device(dev):
            if not torch.cuda.is_current_stream_capturing():
                self._check_aliasing()
                self.sync_hyperparameters()
            stream = vp(torch.cuda.current_stream(dev).cuda_stream)
            for i, b in enumerate(self.reducer.buckets):
                _lib.check(self._lib.pvcnn_adam_step(vp(b.pflat.data_ptr()), vp(b.flat.data_ptr()), vp(self.exp_avg[i].data_ptr()),
                                                     vp(self.exp_avg_sq[i].data_ptr()), b.flat.numel(), vp(self.step_count.data_ptr()),
                                                     vp(self.hyper.data_ptr()), int(i == last), stream), 'adam_step')
        # the kernel writes the parameters through raw pointers: torch's version counters do not move, so tell the holders of
        # derived data (the f16x2 weight images of backend.weight_bank_*) that every parameter changed
        from .modules.functional._autograd import native
        invalidate = getattr(native(), 'weight_bank_invalidate', None)
        if invalidate is not None:
            invalidate()
        return loss

    def zero_grad(self, set_to_none=True):
        self.reducer.zero_grad()

    # ---- checkpoints: torch.optim.Adam's layout (train.py:186-199 saves optimizer.state_dict(), 249-255 restores it) ----
    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..n-1]}]} -- what
        torch.optim.Adam(model.parameters(), capturable=True).state_dict() holds for the same model (tensors are copies)."""
        params = self.param_groups[0]['params']
        state = {i: {'step': self.step_count.detach().clone().view(()), 'exp_avg': self.state[p]['exp_avg'].detach().clone(),
                     'exp_avg_sq': self.state[p]['exp_avg_sq'].detach().clone()} for i, p in enumerate(params)}
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(params)))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam's state_dict of the same model (any `capturable` / `fused` flavour: `step` may be a float or a
        tensor) -- and therefore FlatAdam's own.  The moments are copied INTO the flat buffers (the kernel's addresses never change: a
        captured graph stays valid); a parameter without saved state (never stepped) keeps zero moments."""
        groups = state_dict['param_groups']
        # validate BEFORE any state is touched.  FlatAdam has ONE set of hyper-parameters (one elementwise pass per bucket): a checkpoint
        # with several param groups -- e.g. a no-decay group for BatchNorm / biases -- loads only if they all agree (it used to
        # load silently with groups[0]'s values)
        keys = ('lr', 'betas', 'eps', 'weight_decay')
        for g in groups:
            if g.get('amsgrad') or g.get('maximize'):
                raise ValueError('FlatAdam implements neither amsgrad nor maximize')
            for key in keys:
                a, b = g.get(key), groups[0].get(key)
                if (tuple(a) if key == 'betas' and a is not None else a) != (tuple(b) if key == 'betas' and b is not None else b):
                    raise ValueError(f'FlatAdam keeps one set of hyper-parameters; the checkpoint\'s {len(groups)} param groups differ in {key!r}: '
                                     f'{[gg.get(key) for gg in groups]}')
        ids = [i for g in groups for i in g['params']]
        params = self.param_groups[0]['params']
        if len(ids) != len(params):
            raise ValueError(f'optimizer state for {len(ids)} parameters, this model has {len(params)}')
        for i, p in zip(ids, params):
            st = state_dict['state'].get(i)
            if st is not None and tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state {i}: shape {tuple(st["exp_avg"].shape)} vs parameter {tuple(p.shape)}')
        found = {float(st['step']) for st in (state_dict['state'].get(i) for i in ids) if st is not None}
        if len(found) > 1:
            raise ValueError(f'FlatAdam keeps ONE step counter; the checkpoint has parameters at steps {sorted(found)}')
        steps = set()
        for i, p in zip(ids, params):
            st = state_dict['state'].get(i)
            mine = self.state[p]
            if st is None:
                mine['exp_avg'].zero_()
                mine['exp_avg_sq'].zero_()
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state {i}: shape {tuple(st["exp_avg"].shape)} vs parameter {tuple(p.shape)}')
            mine['exp_avg'].copy_(st['exp_avg'])
            mine['exp_avg_sq'].copy_(st['exp_avg_sq'])
            steps.add(float(st['step']))
        if len(steps) > 1:
            raise ValueError(f'FlatAdam keeps ONE step counter; the checkpoint has parameters at steps {sorted(steps)}')
        self.step_count.fill_(steps.pop() if steps else 0.0)
        g = groups[0]
        for key in ('lr', 'betas', 'eps', 'weight_decay', 'initial_lr'):
            if key in g:
                self.param_groups[0][key] = tuple(g[key]) if key == 'betas' else g[key]
        self.sync_hyperparameters()
