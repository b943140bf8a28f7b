#!/usr/bin/env python3
"""One Conv3d backward-weight launch (f16x2) timed alone, from the product library or from the ablation build of its unit
(tools/probe/libablate_conv3d_wgrad_f16.so with PVCNN_WGRAD_ABLATE=<bits>: 1 no global loads, 2 no conversion / LDS stores, 4 no MFMAs,
8 no partial store, 16 no row barrier).

    python tools/wgrad_only.py [--ablate] [--shapes 16x64x64x16,16x64x64x32]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pvcnn_amd import _lib  # noqa: E402

product = _lib.load()
dev = 'cuda:0'


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    probe = '--probe' in sys.argv
    ablate = '--ablate' in sys.argv or probe
    lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', ('libprobe' if probe else 'libablate') + '_conv3d_wgrad_f16.so')) if ablate else product
    pbuf = torch.zeros(32, dtype=torch.int64, device=dev)
    if probe:
        assert lib.pvcnn_probe_set_buffer(P(pbuf)) == 0
    lib.pvcnn_conv3d_bwd_weight_f16_workspace_bytes.restype = ctypes.c_size_t
    lib.pvcnn_conv3d_bwd_weight_f16.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    shapes = '16x64x64x16,16x64x128x16,16x128x128x16,16x64x64x32,16x9x64x32'
    if '--shapes' in sys.argv:
        shapes = sys.argv[sys.argv.index('--shapes') + 1]
    g = torch.Generator(device=dev).manual_seed(5)
    for shp in shapes.split(','):
        b, ci, co, r = (int(v) for v in shp.split('x'))
        x = torch.randn(b, ci, r, r, r, device=dev, generator=g)
        gy = torch.randn(b, co, r, r, r, device=dev, generator=g)
        ax = torch.empty(product.pvcnn_absmax_tiles_count(b, r ** 3, r), dtype=torch.int32, device=dev)
        ag = torch.empty_like(ax)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(product.pvcnn_absmax_tiles(P(x), b, ci, r ** 3, r, P(ax), None, s), 'absmax')
        _lib.check(product.pvcnn_absmax_tiles(P(gy), b, co, r ** 3, r, P(ag), None, s), 'absmax')
        gw = torch.empty(co, ci, 3, 3, 3, device=dev)
        gb = torch.empty(co, device=dev)
        nb = lib.pvcnn_conv3d_bwd_weight_f16_workspace_bytes(b, ci, co, r)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)

        def launch():
            rc = lib.pvcnn_conv3d_bwd_weight_f16(P(x), P(gy), P(ax), 0, P(ag), 0, b, ci, co, r, P(gw), P(gb), P(ws), nb, s)
            assert rc == 0, rc
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        row = {'BCiCoR': [b, ci, co, r], 'ablate': int(os.environ.get('PVCNN_WGRAD_ABLATE', '0')) if ablate else None, 'two_launches_us': round(best, 1)}
        if not ablate or row['ablate'] == 0:
            xd = x.double()
            ref = torch.zeros(co, ci, 3, 3, 3, dtype=torch.float64, device=dev)
            import torch.nn.functional as F
            wd = torch.zeros(co, ci, 3, 3, 3, dtype=torch.float64, device=dev, requires_grad=True)
            F.conv3d(xd, wd, padding=1).backward(gy.double())
            row['rel_err'] = float((gw.double() - wd.grad).abs().max() / wd.grad.abs().max())
            del ref
        if probe:
            torch.cuda.synchronize()
            pbuf.zero_()
            launch()
            torch.cuda.synchronize()
            sl = pbuf.cpu().tolist()
            names = ['top (requests, live check)', 'convert (if first)', 'multiply', 'convert (if last)', 'barrier']
            for base, who in ((0, 'waves 0-3 (multiply first)'), (8, 'waves 4-7')):
                tot = sum(sl[base:base + 5])
                row[who] = {names[k]: round(sl[base + k] / max(tot, 1), 3) for k in range(5)}
                row[who]['cycles_per_wave'] = round(tot / max(sl[31] / 2, 1))
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
