"""The reference's OWN model composition on pvcnn_amd.modules -- test / bench infrastructure, not part of the product.

`pvcnn_amd.workload` assembles the BASELINE networks with caller-side shortcuts around the operator API (the global max-pool's winners
out of the BatchNorm pass, `expand` instead of `repeat`, a concatenation kernel that also emits the scale table, the classifier head run
module by module with its Dropout fused and its last Conv1d on this package's GEMM).  A user of the reference who only swaps the
`modules` package gets none of those: their `models/` compose the operators with plain torch.  The classes below are that composition,
statement by statement -- what the reference's forward() methods do, on this package's modules and nothing else:

    ReferencePVCNN          models/s3dis/pvcnn.py:34-46      .max(dim=-1).values, .unsqueeze(-1).repeat, torch.cat, self.classifier(...)
    ReferencePVCNN2         models/s3dis/pvcnnpp.py:44-59    the SA / FP loops, self.classifier(...)
    ReferencePVCNNShapeNet  models/shapenet/pvcnn.py:30-42   .max(dim=-1, keepdim=True).values.repeat, torch.cat, self.classifier(...)
    ReferenceFrustumPVCNNE  models/kitti/frustum/frustum_net.py:36-67 on segmentation/pointnet.py:34-44, center_regression_net.py:25-32,
                            box_estimation/pointnet.py:33-40  (.repeat, .max, torch.cat, the heads as plain nn.Sequentials)

with `self.classifier` / `self.cloud_features` the plain `nn.Sequential`s of models/utils.py:15-46 (SharedMLP, nn.Dropout, nn.Conv1d /
Linear + BatchNorm1d + ReLU) called as modules.  The constructors are workload's (same sub-module tree, same state_dict keys as the
reference classes: tests/test_reference_python.py), only forward() differs.  Where /root/reference is mounted,
tests/test_reference_python.py::test_reference_composition_is_the_reference_forward runs these against the reference's own classes
(bit-equal outputs and gradients on the CPU oracle stack); on the GPU tests/test_gpu_reference_composition.py compares them with the
workload classes, and `bench.py --reference-composition` times them.
"""
import torch

from pvcnn_amd import workload

__all__ = ['ReferencePVCNN', 'ReferencePVCNN2', 'ReferencePVCNNShapeNet', 'ReferenceFrustumPVCNNE', 'BY_CONFIG']


class ReferencePVCNN(workload.PVCNN):
    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = inputs['features']
        xyz = inputs[:, :3, :]
        collected = []
        for block in self.point_features:
            inputs, _ = block((inputs, xyz))
            collected.append(inputs)
        descriptor = self.cloud_features(inputs.max(dim=-1, keepdim=False).values)
        collected.append(descriptor.unsqueeze(-1).repeat([1, 1, xyz.size(-1)]))
        return self.classifier(torch.cat(collected, dim=1))


class ReferencePVCNN2(workload.PVCNN2):
    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = inputs['features']
        xyz, feats = inputs[:, :3, :].contiguous(), inputs
        xyz_levels, skip_levels = [], []
        for level in self.sa_layers:
            skip_levels.append(feats)
            xyz_levels.append(xyz)
            feats, xyz = level((feats, xyz))
        skip_levels[0] = inputs[:, 3:, :].contiguous()
        for i, level in enumerate(self.fp_layers):
            feats, xyz = level((xyz_levels[-1 - i], xyz, feats, skip_levels[-1 - i]))
        return self.classifier(feats)


class ReferencePVCNNShapeNet(workload.PVCNNShapeNet):
    def forward(self, inputs):
        feats = inputs[:, :self.in_channels, :]
        shape_code = inputs[:, -self.num_shapes:, :]
        npts = feats.size(-1)
        xyz = feats[:, :3, :]
        collected = [shape_code]
        for block in self.point_features:
            feats, _ = block((feats, xyz))
            collected.append(feats)
        collected.append(feats.max(dim=-1, keepdim=True).values.repeat([1, 1, npts]))
        return self.classifier(torch.cat(collected, dim=1))


class _ReferenceFrustumSegmentation(workload._FrustumSegmentation):
    def forward(self, inputs):
        feats = inputs['features']
        npts = feats.size(-1)
        one_hot = inputs['one_hot_vectors'].unsqueeze(-1).repeat([1, 1, npts])
        per_point, xyz = self.point_features((feats, feats[:, :3, :]))
        pooled, _ = self.cloud_features((per_point, xyz))
        pooled = pooled.max(dim=-1, keepdim=True).values.repeat([1, 1, npts])
        return self.classifier(torch.cat([one_hot, per_point, pooled], dim=1))


class _ReferenceCenterRegression(workload._CloudRegressor):
    def forward(self, inputs):
        desc = self.features(inputs['coords']).max(dim=-1, keepdim=False).values
        return self.regression(torch.cat([desc, inputs['one_hot_vectors']], dim=1))


class _ReferenceBoxEstimation(workload._CloudRegressor):
    def forward(self, inputs):
        xyz = inputs['coords']
        desc, _ = self.features((xyz, xyz))
        desc = desc.max(dim=-1, keepdim=False).values
        return self.classifier(torch.cat([desc, inputs['one_hot_vectors']], dim=1))


class ReferenceFrustumPVCNNE(workload.FrustumPVCNNE):
    """workload's constructor (the reference's sub-module tree and state_dict keys), the three sub-nets with the reference's forward();
    FrustumNet.forward itself (frustum_net.py:36-67) is what workload.FrustumPVCNNE.forward states."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.inst_seg_net.__class__ = _ReferenceFrustumSegmentation
        self.center_reg_net.__class__ = _ReferenceCenterRegression
        self.box_est_net.__class__ = _ReferenceBoxEstimation


# bench.py --reference-composition: BASELINE config -> (class, constructor arguments before width_multiplier)
BY_CONFIG = {'cfg2': (ReferencePVCNN, (13, 6)), 'cfg3': (ReferencePVCNN2, (13, 6)), 'cfg4': (ReferencePVCNNShapeNet, (50, 16, 3))}
