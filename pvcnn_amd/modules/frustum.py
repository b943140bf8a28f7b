"""Frustum-PointNet multi-task loss and box-corner helper (reference: modules/frustum.py:11-124).
(B, <= 8 x 3) tensors: ~260 torch launches per step when written as the reference writes it -- a third of a Frustum-PVCNN step's
launches.  On the GPU path the box part of the loss and its gradient are ONE launch (csrc/frustum.hip: `_BoxLoss` below); the torch
formulation stays as the definition (CPU, other dtypes, double backward) and as the checker of tests/test_gpu_frustum_loss.py."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as tf

from . import functional as PF

__all__ = ['FrustumPointNetLoss', 'get_box_corners_3d']

_GRAD_ORDER = ('center', 'center_reg', 'heading_scores', 'size_scores', 'heading_residuals_normalized', 'size_residuals_normalized',
               'heading_residuals', 'size_residuals')


class _BoxLoss(torch.autograd.Function):
    """(the eight network outputs of _GRAD_ORDER, targets, constants) -> box loss (0-dim); the kernel that evaluates it also writes
    its gradient, backward scales that by the incoming gradient (one elementwise launch)."""

    @staticmethod
    def forward(ctx, *args):
        outs, rest = args[:8], args[8:]
        from .functional._autograd import native
        loss, grads = native().frustum_box_loss(*[t.contiguous() for t in outs], *rest)
        ctx.save_for_backward(grads)
        ctx.shapes = [tuple(t.shape) for t in outs]
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (grads,) = ctx.saved_tensors
        scaled = grads * g
        res, off = [], 0
        for shape in ctx.shapes:
            n = 1
            for d in shape:
                n *= d
            res.append(scaled[off:off + n].view(shape))
            off += n
        return (*res, *([None] * 11))

# corner sign pattern (l, h, w) of the 8 box corners, counter-clockwise, top face first
_SX = (1, 1, -1, -1, 1, 1, -1, -1)
_SY = (1, 1, 1, 1, -1, -1, -1, -1)
_SZ = (1, -1, -1, 1, 1, -1, -1, 1)


_SIGNS = {}      # (device, dtype) -> the three sign rows as device tensors


def _corner_signs(like):
    """The constant sign patterns on `like`'s device: uploaded once per (device, dtype) -- a host-to-device copy inside every
    forward would also be illegal while the step is being captured into a hipGraph (pvcnn_amd/graph.py)."""
    key = (like.device, like.dtype)
    if key not in _SIGNS:
        _SIGNS[key] = tuple(torch.tensor(v, device=like.device, dtype=like.dtype) for v in (_SX, _SY, _SZ))
    return _SIGNS[key]


def get_box_corners_3d(centers, headings, sizes, with_flip=False):
    """centers (N,3), headings (N,), sizes (N,3)=(l,w,h) -> corners (N,3,8) rotated about y;
    with_flip also returns the box turned by pi."""
    half = sizes / 2
    sx, sy, sz = _corner_signs(half)
    local = torch.stack([half[:, 0:1] * sx, half[:, 2:3] * sy, half[:, 1:2] * sz], dim=1)   # (N,3,8)
    c, s = torch.cos(headings), torch.sin(headings)
    o, z = torch.ones_like(headings), torch.zeros_like(headings)
    shift = centers.unsqueeze(-1)
    rot = torch.stack([c, z, s, z, o, z, -s, z, c], dim=1).view(-1, 3, 3)
    if not with_flip:
        return torch.matmul(rot, local) + shift
    rot_flip = torch.stack([-c, z, -s, z, o, z, s, z, -c], dim=1).view(-1, 3, 3)
    return torch.matmul(rot, local) + shift, torch.matmul(rot_flip, local) + shift


class FrustumPointNetLoss(nn.Module):
    def __init__(self, num_heading_angle_bins, num_size_templates, size_templates, box_loss_weight=1.0,
                 corners_loss_weight=10.0, heading_residual_loss_weight=20.0, size_residual_loss_weight=20.0):
        super().__init__()
        self.box_loss_weight = box_loss_weight
        self.corners_loss_weight = corners_loss_weight
        self.heading_residual_loss_weight = heading_residual_loss_weight
        self.size_residual_loss_weight = size_residual_loss_weight
        self.num_heading_angle_bins = num_heading_angle_bins
        self.num_size_templates = num_size_templates
        self.register_buffer('size_templates', size_templates.view(self.num_size_templates, 3))
        self.register_buffer('heading_angle_bin_centers',
                             torch.arange(0, 2 * math.pi, 2 * math.pi / self.num_heading_angle_bins))

    def forward(self, inputs, targets):
        # fp32 under torch.autocast as well: the loss is a few hundred scalars -- the 3x3 rotations of the corner loss (torch.matmul)
        # would otherwise run on bf16 operands, with a cast kernel per operand (BASELINE configs[4] asks for bf16 in the dense
        # convolutions, not here)
        up = lambda t: t.float() if t.dtype in (torch.bfloat16, torch.float16) else t
        inputs = {k: up(v) for k, v in inputs.items()}
        with torch.autocast(inputs['center'].device.type, enabled=False):
            return self._forward(inputs, targets)

    def _fused_ok(self, inputs, targets):
        if not inputs['center'].is_cuda or os.environ.get('PVCNN_FUSED_FRUSTUM_LOSS', '1') == '0':      # (0: debug / A-B, the torch formulation)
            return False
        from .functional._autograd import native
        if not getattr(native(), 'has_frustum_loss', False):
            return False
        f32 = all(inputs[k].dtype == torch.float32 and inputs[k].is_cuda for k in _GRAD_ORDER)
        tg = all(targets[k].dtype == torch.float32 and targets[k].is_cuda for k in ('heading_residual', 'size_residual', 'center'))
        ids = all(targets[k].dtype == torch.int64 and targets[k].is_cuda and targets[k].dim() == 1 for k in ('heading_bin_id', 'size_template_id'))
        return f32 and tg and ids and self.size_templates.dtype == torch.float32 and self.size_templates.is_cuda

    def _forward(self, inputs, targets):
        if self._fused_ok(inputs, targets):
            # csrc/frustum.hip: every term below except the mask's cross entropy, and its gradient, in one launch
            box = _BoxLoss.apply(*[inputs[k] for k in _GRAD_ORDER], targets['heading_bin_id'].contiguous(), targets['size_template_id'].contiguous(),
                                 targets['heading_residual'].contiguous(), targets['size_residual'].contiguous(), targets['center'].contiguous(),
                                 self.size_templates, self.heading_angle_bin_centers, math.pi / self.num_heading_angle_bins,
                                 self.heading_residual_loss_weight, self.size_residual_loss_weight, self.corners_loss_weight)
            return tf.cross_entropy(inputs['mask_logits'], targets['mask_logits']) + self.box_loss_weight * box
        return self._forward_torch(inputs, targets)

    def _forward_torch(self, inputs, targets):
        center = inputs['center']
        rows = torch.arange(center.size(0), device=center.device)
        h_id, s_id = targets['heading_bin_id'], targets['size_template_id']
        h_res_t, s_res_t, center_t = targets['heading_residual'], targets['size_residual'], targets['center']
        bin_centers, templates = self.heading_angle_bin_centers, self.size_templates

        # classification / coarse regression
        cls = (tf.cross_entropy(inputs['heading_scores'], h_id) + tf.cross_entropy(inputs['size_scores'], s_id))
        mask_loss = tf.cross_entropy(inputs['mask_logits'], targets['mask_logits'])
        center_loss = PF.huber_loss(torch.norm(center_t - center, dim=-1), delta=2.0)
        center_reg_loss = PF.huber_loss(torch.norm(center_t - inputs['center_reg'], dim=-1), delta=1.0)

        # normalised residuals of the target bin / template
        h_norm_loss = PF.huber_loss(
            inputs['heading_residuals_normalized'][rows, h_id] - h_res_t / (math.pi / self.num_heading_angle_bins),
            delta=1.0)
        s_norm_loss = PF.huber_loss(
            torch.norm(s_res_t / templates[s_id] - inputs['size_residuals_normalized'][rows, s_id], dim=-1), delta=1.0)

        # corner loss against the target box and its pi-flipped twin
        heading = inputs['heading_residuals'][rows, h_id] + bin_centers[h_id]
        size = inputs['size_residuals'][rows, s_id] + templates[s_id]
        corners = get_box_corners_3d(centers=center, headings=heading, sizes=size, with_flip=False)
        tgt, tgt_flip = get_box_corners_3d(centers=center_t, headings=bin_centers[h_id] + h_res_t,
                                           sizes=templates[s_id] + s_res_t, with_flip=True)
        corners_loss = PF.huber_loss(
            torch.min(torch.norm(corners - tgt, dim=1), torch.norm(corners - tgt_flip, dim=1)), delta=1.0)

        box = (center_loss + center_reg_loss + cls
               + self.heading_residual_loss_weight * h_norm_loss
               + self.size_residual_loss_weight * s_norm_loss
               + self.corners_loss_weight * corners_loss)
        return mask_loss + self.box_loss_weight * box
