#!/usr/bin/env python3
"""The small 1x1 layers (K, M <= 128 over 65 536 points) one launch at a time: forward GEMM, backward-data, backward-weight (fp32-MFMA
"small" kernel + its reduce), graph-replayed back to back.  usage: python tools/pw_small_bench.py [--shapes BxKxMxN,...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pvcnn_amd.modules.functional.backend import HipBackend  # noqa: E402

be = HipBackend()
dev = 'cuda:0'


def graph_time(fn, reps=20, iters=5):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    shapes = '16x64x64x4096,16x64x128x4096,16x128x128x4096,8x32x64x8192,8x64x64x2048'
    if '--shapes' in sys.argv:
        shapes = sys.argv[sys.argv.index('--shapes') + 1]
    for shp in shapes.split(','):
        b, k, m, n = (int(v) for v in shp.split('x'))
        x, gy = torch.randn(b, k, n, device=dev), torch.randn(b, m, n, device=dev)
        w = torch.randn(m, k, device=dev) * 0.1
        row = {'BKMN': [b, k, m, n], 'PMAX': os.environ.get('PVCNN_PW_SMALL_PMAX', 'default')}
        gw = be.pwconv_backward_weight(x, gy)
        ref = torch.einsum('bmn,bkn->mk', gy.double(), x.double())
        row['wgrad_rel_err'] = float((gw.double() - ref).abs().max() / ref.abs().max())
        row['wgrad_us'] = round(graph_time(lambda: be.pwconv_backward_weight(x, gy)), 1)
        row['wgrad_with_bias_us'] = round(graph_time(lambda: be.pwconv_backward_weight(x, gy, with_bias=True)), 1)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
