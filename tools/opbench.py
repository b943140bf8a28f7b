#!/usr/bin/env python3
"""Op-level microbenchmark of the hand-written kernels at SURVEY.md 8(d)'s op-level shapes.

Prints one JSON line per (op, shape): median / min launch time from HIP events on the launch
stream, algorithmic GB/s and fraction of the 8 TB/s HBM peak.  Used for the rocprofv3 runs whose
summaries live in profiles/ and for within-run A/B comparisons while tuning.
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench as B  # byte formulas

SHAPES = [(16, 64, 4096, 32), (16, 64, 4096, 16), (16, 128, 4096, 16), (64, 128, 2048, 16), (8, 32, 8192, 32),
          (32, 64, 1024, 12), (16, 9, 4096, 32)]


def timeit(fn, iters, warm=3, reps=10):
    """Per-call device time: `reps` back-to-back calls captured in a hipGraph (host launch overhead
    -- ~20 us of Python/ctypes per call -- is outside the measurement), replayed `iters` times."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--ops', default='vox_fwd,vox_apply,vox_plan,vox_bwd,devox_fwd,devox_bwd,devox_bwd_apply,devox_bwd_plan')
    ap.add_argument('--shapes', default='')
    ap.add_argument('--kind', default='cube', choices=['cube', 'surface'])
    ap.add_argument('--no-seam-memo', action='store_true',
                    help='vox_fwd / devox_bwd (the 12-callable seam): the one-shot C entries, counting sort rebuilt per call (the FIRST call on a '
                         'tensor); default: the memoised plan of the tensor they are called with (every later call)')
    args = ap.parse_args()
    from pvcnn_amd.modules.functional.backend import _backend as hip
    hip.seam_plan_memo = not args.no_seam_memo
    dev = 'cuda:0'
    shapes = SHAPES if not args.shapes else [tuple(int(v) for v in s.split('x')) for s in args.shapes.split(',')]
    g = torch.Generator().manual_seed(1588147245)
    for (b, c, n, r) in shapes:
        s = r ** 3
        co = torch.rand(b, 3, n, generator=g)
        if args.kind == 'surface':
            plane = torch.randint(0, 3, (b, n), generator=g)
            for ax in range(3):
                co[:, ax, :] = torch.where(plane == ax, torch.full_like(co[:, ax, :], 0.5), co[:, ax, :])
        norm = torch.clamp(co * r, 0, r - 1).contiguous().to(dev)
        vox = torch.round(norm).to(torch.int32).contiguous()
        feat = torch.randn(b, c, n, generator=g).to(dev)
        grid = torch.randn(b, c, s, generator=g).to(dev)
        out, ind, cnt = hip.avg_voxelize_forward(feat, vox, r)
        outs, inds, wgts = hip.trilinear_devoxelize_forward(r, True, norm, grid)
        gy_pts = torch.randn(b, c, n, generator=g).to(dev)
        vplan = hip.avg_voxelize_plan(vox, r)
        dplan = hip.trilinear_devoxelize_backward_plan(inds, wgts, r)
        runs = {
            # plan / apply split (what a network runs: one plan per (coords, R), one apply per layer)
            'vox_plan': (lambda: hip.avg_voxelize_plan(vox, r), 4 * b * (3 * n + n + s)),
            'vox_apply': (lambda: hip.avg_voxelize_apply(feat, vplan), B.bytes_vox_fwd(b, c, n, s)),
            'devox_bwd_plan': (lambda: hip.trilinear_devoxelize_backward_plan(inds, wgts, r), 4 * b * 16 * n),
            'devox_bwd_apply': (lambda: hip.trilinear_devoxelize_backward_apply(gy_pts, dplan, r), B.bytes_devox_bwd(b, c, n, s)),
            'vox_fwd': (lambda: hip.avg_voxelize_forward(feat, vox, r), B.bytes_vox_fwd(b, c, n, s)),
            'vox_bwd': (lambda: hip.avg_voxelize_backward(grid, ind, cnt), B.bytes_vox_bwd(b, c, n, s)),
            'devox_fwd': (lambda: hip.trilinear_devoxelize_forward(r, True, norm, grid), B.bytes_devox_fwd(b, c, n, s, True)),
            'devox_fwd_eval': (lambda: hip.trilinear_devoxelize_forward(r, False, norm, grid), B.bytes_devox_fwd(b, c, n, s, False)),
            'devox_bwd': (lambda: hip.trilinear_devoxelize_backward(gy_pts, inds, wgts, r), B.bytes_devox_bwd(b, c, n, s)),
        }
        for op in args.ops.split(','):
            fn, nbytes = runs[op]
            med, best = timeit(fn, args.iters)
            print(json.dumps({'op': op + (' (one-shot, plan rebuilt)' if args.no_seam_memo and op in ('vox_fwd', 'devox_bwd') else ''), 'BCNR': [b, c, n, r], 'kind': args.kind, 'median_us': round(med, 2), 'min_us': round(best, 2),
                              'algorithmic_MB': round(nbytes / 1e6, 2), 'GBs': round(nbytes / med / 1e3, 1),
                              'frac_8TBs': round(nbytes / med / 1e3 / 8000, 4)}), flush=True)


if __name__ == '__main__':
    main()
