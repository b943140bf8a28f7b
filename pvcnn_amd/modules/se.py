"""SE3d: squeeze-and-excitation over a voxel grid (reference: modules/se.py:6-17).
x * sigmoid(W2 relu(W1 mean_xyz(x))), reduction 8, bias-free; `fc.0` / `fc.2` hold the weights."""
import torch.nn as nn

__all__ = ['SE3d']


class SE3d(nn.Module):
    def __init__(self, channel, reduction=8):
        super().__init__()
        hidden = channel // reduction
        self.fc = nn.Sequential(nn.Linear(channel, hidden, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(hidden, channel, bias=False), nn.Sigmoid())

    def forward(self, inputs):
        nb, nc = inputs.shape[0], inputs.shape[1]
        # mean over z, then y, then x -- the reference's reduction order (se.py:17)
        squeezed = inputs.mean(-1).mean(-1).mean(-1)
        return inputs * self.fc(squeezed).view(nb, nc, 1, 1, 1)
