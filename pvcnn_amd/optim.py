"""Adam on the flat buckets of the data-parallel step (csrc/optim.hip).

Reference: train.py:96-119 calls `optimizer.step()` of torch.optim.Adam on the model's ~100 parameter tensors.  Here the parameters
are laid out like the gradient buckets of `pvcnn_amd.dp.GradBucketReducer` (flatten_parameters) and one elementwise kernel updates a
whole bucket: same arithmetic as torch.optim.Adam (no amsgrad; weight decay added to the gradient), fp32, step counter on the device,
so the update is graph-capturable (pvcnn_amd/graph.py) and costs ~10 us instead of 2 x 72 us for PVCNN's 9.8 MiB.

GPU only (there is no CPU implementation of the kernel); use torch.optim.Adam for CPU runs.
"""
import ctypes

import torch

from . import _lib

__all__ = ['FlatAdam']


class FlatAdam:
    """opt = FlatAdam(reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0);  per step, after reducer.finish(): opt.step().

    The gradients are read from the reducer's flat buckets (where `finish()` leaves them: `p.grad` are views of those buffers)."""

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not reducer.buckets or not all(b.flat.is_cuda and b.flat.dtype == torch.float32 for b in reducer.buckets):
            raise ValueError('FlatAdam needs float32 parameters on a GPU (use torch.optim.Adam elsewhere)')
        reducer.flatten_parameters()
        self.reducer = reducer
        self.defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay),
                             capturable=True)
        self.param_groups = [dict(self.defaults, params=list(reducer.params))]
        self.exp_avg = [torch.zeros_like(b.flat) for b in reducer.buckets]
        self.exp_avg_sq = [torch.zeros_like(b.flat) for b in reducer.buckets]
        self.step_count = torch.zeros((1,), dtype=torch.float32, device=reducer.buckets[0].flat.device)
        self._lib = _lib.load()

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        vp = ctypes.c_void_p
        last = len(self.reducer.buckets) - 1
        dev = self.step_count.device
        with torch.cuda.device(dev):
            stream = vp(torch.cuda.current_stream(dev).cuda_stream)
            for i, b in enumerate(self.reducer.buckets):
                _lib.check(self._lib.pvcnn_adam_step(vp(b.pflat.data_ptr()), vp(b.flat.data_ptr()), vp(self.exp_avg[i].data_ptr()),
                                                     vp(self.exp_avg_sq[i].data_ptr()), b.flat.numel(), vp(self.step_count.data_ptr()),
                                                     g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], int(i == last),
                                                     stream), 'adam_step')

    def zero_grad(self, set_to_none=True):
        self.reducer.zero_grad()

    def state_dict(self):
        return {'step': self.step_count.clone(), 'exp_avg': [t.clone() for t in self.exp_avg], 'exp_avg_sq': [t.clone() for t in self.exp_avg_sq],
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, state):
        self.step_count.copy_(state['step'])
        for dst, src in zip(self.exp_avg, state['exp_avg']):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, state['exp_avg_sq']):
            dst.copy_(src)
        self.param_groups[0].update(state['param_groups'][0])
