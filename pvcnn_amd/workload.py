"""Benchmark workloads: the reference networks assembled from pvcnn_amd.modules + synthetic inputs.

The reference's `models/` are callers of the hot path and are NOT part of this package: with
`pvcnn_amd.install_dropin()` they run unchanged on top of `pvcnn_amd.modules`.  bench.py and
smoke() however run on a GPU box where the reference tree does not exist, so the two S3DIS
networks BASELINE.json names are assembled here from a small declarative spec.  Parameter
names match the reference classes (models/s3dis/pvcnn.py:9-46, models/s3dis/pvcnnpp.py:8-59,
builders models/utils.py:15-140), so `state_dict`s are interchangeable;
tests/test_reference_python.py checks keys, shapes and outputs against the reference itself.

Synthetic inputs follow SURVEY.md 8(d): seed 1588147245 (configs/__init__.py:3), channel
layout of datasets/s3dis.py:90 (block-local xyz in metres, rgb, room-normalised xyz), ~5 % exact
duplicate points (the loader samples with replacement when a window holds < N points).
"""
import torch
import torch.nn as nn

from .modules import PVConv, PointNetAModule, PointNetFPModule, PointNetSAModule, SharedMLP

SEED = 1588147245

__all__ = ['PVCNN', 'PVCNN2', 'make_s3dis_batch', 'SEED']


def _scaled(width, k):
    return int(k * width)


def _dense_bn_relu(cin, cout):
    return nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.ReLU(True))


def _head(cin, spec, width, pointwise, classify):
    """MLP head from a spec like [512, 0.3, 256, 0.3, num_classes]: floats < 1 are dropout rates.
    pointwise=True -> SharedMLP / Conv1d on (B,C,N); False -> Linear+BN1d+ReLU on (B,C)."""
    spec = list(spec) if isinstance(spec, (list, tuple)) else [spec]
    block = SharedMLP if pointwise else _dense_bn_relu
    layers = []
    for item in spec[:-1]:
        if item < 1:
            layers.append(nn.Dropout(item))
        else:
            layers.append(block(cin, _scaled(item, width)))
            cin = _scaled(item, width)
    last = spec[-1]
    if classify:
        layers.append(nn.Conv1d(cin, last, 1) if pointwise else nn.Linear(cin, last))
        return layers, last
    layers.append(block(cin, _scaled(last, width)))
    return layers, _scaled(last, width)


def _pv_stack(cin, spec, width, vres, **pvconv_kw):
    """(out_channels, num_blocks, voxel_resolution | None) -> list of PVConv / SharedMLP blocks."""
    cout, repeat, res = spec
    cout = _scaled(cout, width)
    blocks = []
    for _ in range(repeat):
        if res is None:
            blocks.append(SharedMLP(cin, cout))
        else:
            blocks.append(PVConv(cin, cout, kernel_size=3, resolution=int(vres * res), **pvconv_kw))
        cin = cout
    return blocks, cout


class PVCNN(nn.Module):
    """PVCNN for S3DIS semantic segmentation: 4 PVConv + 1 SharedMLP point stages, a global
    max-pooled cloud descriptor, and a point-wise classifier over the concatenation."""
    blocks = ((64, 1, 32), (64, 2, 16), (128, 1, 16), (1024, 1, None))

    def __init__(self, num_classes, extra_feature_channels=6, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        self.in_channels = extra_feature_channels + 3
        stages, cin, concat = [], self.in_channels, 0
        for spec in self.blocks:
            blocks, cin = _pv_stack(cin, spec, width_multiplier, voxel_resolution_multiplier,
                                    with_se=False, normalize=True, eps=0)
            stages += blocks
            concat += cin * len(blocks)
        self.point_features = nn.ModuleList(stages)
        layers, c_cloud = _head(cin, [256, 128], width_multiplier, pointwise=False, classify=False)
        self.cloud_features = nn.Sequential(*layers)
        layers, _ = _head(concat + c_cloud, [512, 0.3, 256, 0.3, num_classes], width_multiplier,
                          pointwise=True, classify=True)
        self.classifier = nn.Sequential(*layers)

    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = inputs['features']
        coords = inputs[:, :3, :]
        feats, taps = inputs, []
        for stage in self.point_features:
            feats, _ = stage((feats, coords))
            taps.append(feats)
        cloud = self.cloud_features(feats.max(dim=-1, keepdim=False).values)
        taps.append(cloud.unsqueeze(-1).repeat([1, 1, coords.size(-1)]))
        return self.classifier(torch.cat(taps, dim=1))


class PVCNN2(nn.Module):
    """PVCNN++ for S3DIS: PointNet++-style set-abstraction / feature-propagation pyramid whose
    stages are PVConv stacks (ball_query / grouping / FPS / 3-NN interpolation path)."""
    sa_blocks = [
        ((32, 2, 32), (1024, 0.1, 32, (32, 64))),
        ((64, 3, 16), (256, 0.2, 32, (64, 128))),
        ((128, 3, 8), (64, 0.4, 32, (128, 256))),
        (None, (16, 0.8, 32, (256, 256, 512))),
    ]
    fp_blocks = [
        ((256, 256), (256, 1, 8)),
        ((256, 256), (256, 1, 8)),
        ((256, 128), (128, 2, 16)),
        ((128, 128, 64), (64, 1, 32)),
    ]

    def __init__(self, num_classes, extra_feature_channels=6, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        self.in_channels = extra_feature_channels + 3
        k, vr = width_multiplier, voxel_resolution_multiplier
        pv_kw = dict(with_se=True, normalize=True, eps=0)

        # --- set abstraction ----------------------------------------------------------------
        sa_layers, skip_channels = [], []
        extra = extra_feature_channels          # channels handed to the SA module besides xyz
        cin = extra_feature_channels + 3        # channels entering the stage's PVConv stack
        for conv_spec, (n_centers, radius, n_nbrs, widths) in self.sa_blocks:
            skip_channels.append(cin)
            stage = []
            if conv_spec is not None:
                blocks, cin = _pv_stack(cin, conv_spec, k, vr, **pv_kw)
                stage += blocks
                extra = cin
            widths = [[_scaled(w, k) for w in ws] if isinstance(ws, (list, tuple)) else _scaled(ws, k) for ws in widths]
            if n_centers is None:
                sa = PointNetAModule(in_channels=extra, out_channels=widths, include_coordinates=True)
            else:
                sa = PointNetSAModule(num_centers=n_centers, radius=radius, num_neighbors=n_nbrs,
                                      in_channels=extra, out_channels=widths, include_coordinates=True)
            stage.append(sa)
            cin = extra = sa.out_channels
            sa_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
        self.sa_layers = nn.ModuleList(sa_layers)

        # --- feature propagation (raw xyz is dropped from the last skip: pvcnnpp.py:38) -------
        skip_channels[0] = extra_feature_channels
        fp_layers = []
        for i, (fp_widths, conv_spec) in enumerate(self.fp_blocks):
            widths = tuple(_scaled(w, k) for w in fp_widths)
            stage = [PointNetFPModule(in_channels=cin + skip_channels[-1 - i], out_channels=widths)]
            cin = widths[-1]
            if conv_spec is not None:
                blocks, cin = _pv_stack(cin, conv_spec, k, vr, **pv_kw)
                stage += blocks
            fp_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
        self.fp_layers = nn.ModuleList(fp_layers)

        layers, _ = _head(cin, [128, 0.5, num_classes], k, pointwise=True, classify=True)
        self.classifier = nn.Sequential(*layers)

    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = inputs['features']
        coords, feats = inputs[:, :3, :].contiguous(), inputs
        coords_pyramid, skips = [], []
        for stage in self.sa_layers:
            skips.append(feats)
            coords_pyramid.append(coords)
            feats, coords = stage((feats, coords))
        skips[0] = inputs[:, 3:, :].contiguous()
        for i, stage in enumerate(self.fp_layers):
            feats, coords = stage((coords_pyramid[-1 - i], coords, feats, skips[-1 - i]))
        return self.classifier(feats)


def make_s3dis_batch(batch, num_points, num_classes=13, device='cpu', seed=SEED, duplicates=0.05):
    """Synthetic S3DIS-like batch: features (B,9,N) fp32 and labels (B,N) int64."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(batch, 3, num_points, generator=g) * torch.tensor([1.5, 1.5, 3.0]).view(1, 3, 1)
    rgb = torch.rand(batch, 3, num_points, generator=g)
    room = torch.rand(batch, 3, num_points, generator=g)
    feats = torch.cat([xyz, rgb, room], dim=1)
    ndup = int(num_points * duplicates)
    if ndup:
        src = torch.randint(0, num_points, (batch, ndup), generator=g)
        dst = torch.randint(0, num_points, (batch, ndup), generator=g)
        for b in range(batch):
            feats[b, :, dst[b]] = feats[b, :, src[b]]
    labels = torch.randint(0, num_classes, (batch, num_points), generator=g)
    return feats.contiguous().to(device), labels.to(device)
