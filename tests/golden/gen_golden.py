#!/usr/bin/env python3
"""Generate tests/golden/*.pt from the REFERENCE ITSELF, run in the build container.

Producer of every vector here:
  * native ops  : oracle/_ref/libpvcnn_ref_cpu.so = the reference's own .cu kernels
                  (/root/reference/modules/functional/src/**) executed on the CPU (oracle/build_ref.py);
  * module level: the reference's own Python (`modules.PVConv`, `models.s3dis.PVCNN`) imported from
                  /root/reference with those kernels plugged in at its `_backend` seam.
Nothing of pvcnn_amd or of the oracle restatement takes part in producing the expected values.
Sizes are powers of two <= 512 wherever the reference accumulates with atomicAdd, so the expected
values are independent of the CUDA schedule (see tests/test_oracle_vs_ref.py).
Run:  python tests/golden/gen_golden.py      (needs /root/reference; rewrites the .pt files)
"""
import importlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from conftest import SEED, grid_coords, synth_cloud   # noqa: E402
from oracle import build_ref, ref_backend             # noqa: E402


def main():
    build_ref.main()
    ref = ref_backend.RefCpuBackend('fma')
    g = torch.Generator().manual_seed(SEED)
    out = {}

    b, c, n, r = 2, 5, 256, 8
    feat = torch.randn(b, c, n, generator=g)
    vox = torch.randint(0, r, (b, 3, n), generator=g, dtype=torch.int32)
    vox[:, :, :64] = vox[:, :, 64:128]
    o, ind, cnt = ref.avg_voxelize_forward(feat, vox.contiguous(), r)
    gy = torch.randn(b, c, r ** 3, generator=g)
    out['avg_voxelize'] = dict(feat=feat, coords=vox, r=r, out=o, ind=ind, cnt=cnt, grad_y=gy,
                               grad_x=ref.avg_voxelize_backward(gy, ind, cnt))

    for name, (b, c, n, r) in {'devox_r8': (2, 5, 512, 8), 'devox_r12': (1, 3, 256, 12)}.items():
        grid = torch.randn(b, c, r ** 3, generator=g)
        co = grid_coords(g, b, n, r)
        outs, inds, wgts = ref.trilinear_devoxelize_forward(r, True, co, grid)
        gp = torch.randn(b, c, n, generator=g)
        out[name] = dict(grid=grid, coords=co, r=r, outs=outs, inds=inds, wgts=wgts, grad_y=gp,
                         grad_x=ref.trilinear_devoxelize_backward(gp, inds, wgts, r),
                         outs_eval=ref.trilinear_devoxelize_forward(r, False, co, grid)[0])

    b, n, m, u = 2, 1024, 128, 16
    pts = synth_cloud(g, b, n, 's3dis')
    fps_idx = ref.furthest_point_sampling(pts, m)
    ctr = ref.gather_features_forward(pts, fps_idx)
    nbr = ref.ball_query(ctr, pts, 0.25, u)
    f = torch.randn(b, 4, n, generator=g)
    grouped = ref.grouping_forward(f, nbr)
    out['sa_stage'] = dict(points=pts, m=m, fps_idx=fps_idx, centers=ctr, radius=0.25, u=u, nbr=nbr, feat=f, grouped=grouped)
    # grouping / gather backward with one (channel, centre) pair per reference thread -> schedule-free
    f2 = torch.randn(2, 4, 300, generator=g)
    idx2 = torch.randint(0, 300, (2, 32, 8), generator=g, dtype=torch.int32)
    g2 = torch.randn(2, 4, 32, 8, generator=g)
    gi = torch.randint(0, 300, (2, 64), generator=g, dtype=torch.int32)
    gg = torch.randn(2, 4, 64, generator=g)
    out['scatter_bwd'] = dict(n=300, idx=idx2, grad_grouped=g2, grad_x_grouping=ref.grouping_backward(g2, idx2, 300),
                              gidx=gi, grad_gathered=gg, grad_x_gather=ref.gather_features_backward(gg, gi, 300))

    b, c, m, n = 2, 4, 40, 64
    p2 = synth_cloud(g, b, n, 's3dis')
    c2 = p2[:, :, torch.randperm(n, generator=g)[:m]].contiguous()
    cf = torch.randn(b, c, m, generator=g)
    o3, i3, w3 = ref.three_nearest_neighbors_interpolate_forward(p2, c2, cf)
    g3 = torch.randn(b, c, n, generator=g)
    out['three_nn'] = dict(points=p2, centers=c2, feat=cf, out=o3, idx=i3, w=w3, grad_y=g3,
                           grad_x=ref.three_nearest_neighbors_interpolate_backward(g3, i3, w3, m))

    # lattice FPS: every step is decided by the reference's tie rule (N = 729 > 512 slots)
    ax = torch.arange(9, dtype=torch.float32)
    lat = torch.stack(torch.meshgrid(ax, ax, ax, indexing='ij')).reshape(1, 3, -1).contiguous()
    out['fps_lattice'] = dict(points=lat, m=60, idx=ref.furthest_point_sampling(lat, 60))

    # ---- module level: the reference's own Python on top of its own kernels -----------------------
    fake = types.ModuleType('modules.functional.backend')
    fake._backend = ref
    sys.modules['modules.functional.backend'] = fake
    sys.path.insert(0, '/root/reference')
    ref_modules = importlib.import_module('modules')
    ref_models = importlib.import_module('models.s3dis')
    torch.manual_seed(SEED)
    layer = ref_modules.PVConv(9, 16, 3, 8, with_se=True, normalize=True).eval()
    x = torch.rand(2, 9, 256, generator=g)
    with torch.no_grad():
        y, _ = layer((x, x[:, :3, :]))
    out['pvconv_eval'] = dict(state=layer.state_dict(), x=x, y=y, ctor=dict(in_channels=9, out_channels=16, kernel_size=3,
                                                                             resolution=8, with_se=True, normalize=True))
    net = ref_models.PVCNN(13, 6, width_multiplier=0.125).eval()
    xin = torch.rand(1, 9, 512, generator=g)
    with torch.no_grad():
        logits = net(xin)
    out['pvcnn_c0p125_eval'] = dict(state=net.state_dict(), x=xin, logits=logits)

    for name, blob in out.items():
        torch.save(blob, os.path.join(HERE, f'{name}.pt'))
        print(name, os.path.getsize(os.path.join(HERE, f'{name}.pt')), 'bytes')


if __name__ == '__main__':
    main()
