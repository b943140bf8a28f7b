"""Shared pieces of the autograd glue: the late-bound native seam and the AMP boundary.

The native ops are fp32-only (like the reference: CHECK_IS_FLOAT, utils.hpp:16-18).  Under
torch.autocast (BASELINE config 5 runs the 3-D convolutions in bf16) floating inputs are cast
to fp32 at the op boundary by custom_fwd and gradients come back in fp32.
"""
import torch

from . import backend as _be


def native():
    """The active `_backend` object, looked up at call time (tests swap it at this one seam)."""
    return _be._backend


amp_fwd = torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
amp_bwd = torch.amp.custom_bwd(device_type='cuda')
