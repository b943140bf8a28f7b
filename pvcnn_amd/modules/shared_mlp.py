"""SharedMLP: a stack of 1x1 Conv{1,2}d + BatchNorm + ReLU applied point-wise.

Reference: modules/shared_mlp.py:6-33.  The layers live in `self.layers` as a flat
nn.Sequential [conv, bn, relu, conv, bn, relu, ...] so checkpoint keys
(`layers.0.weight`, `layers.1.running_mean`, ...) are the reference's.  A tuple / list input
is treated as (features, *extras): only the features pass through the MLP."""
import torch.nn as nn

from .functional.bnact import run_layers

__all__ = ['SharedMLP']

_BY_DIM = {1: (nn.Conv1d, nn.BatchNorm1d), 2: (nn.Conv2d, nn.BatchNorm2d)}


class SharedMLP(nn.Module):
    def __init__(self, in_channels, out_channels, dim=1):
        super().__init__()
        if dim not in _BY_DIM:
            raise ValueError
        conv, norm = _BY_DIM[dim]
        widths = list(out_channels) if isinstance(out_channels, (list, tuple)) else [out_channels]
        stack, cin = [], in_channels
        for cout in widths:
            stack += [conv(cin, cout, 1), norm(cout), nn.ReLU(True)]
            cin = cout
        self.layers = nn.Sequential(*stack)

    def forward(self, inputs):
        # run_layers == self.layers(x) with each BatchNorm + ReLU pair fused into two passes on the GPU
        if isinstance(inputs, (list, tuple)):
            head, *rest = inputs
            return (run_layers(self.layers, head), *rest)
        return run_layers(self.layers, inputs)
