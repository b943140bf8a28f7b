#!/usr/bin/env python3
"""Correctness (vs an fp64 torch convolution) and speed of the MFMA Conv3d kernels, raw C-ABI calls."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from pvcnn_amd import _lib

lib = _lib.load()
dev = 'cuda:0'


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def fwd(x, w, bias):
    b, ci, r = x.shape[0], x.shape[1], x.shape[2]
    co = w.shape[0]
    wt = torch.empty(ci * 27 * co, device=dev)
    y = torch.empty(b, co, r, r, r, device=dev)
    _lib.check(lib.pvcnn_conv3d_weight_transform(P(w), co, ci, 0, P(wt), S()), 'wt')
    _lib.check(lib.pvcnn_conv3d_fwd(P(x), P(wt), P(bias), b, ci, co, r, P(y), S()), 'fwd')
    return y


def bwd_data(gy, w):
    b, co, r = gy.shape[0], gy.shape[1], gy.shape[2]
    ci = w.shape[1]
    wt = torch.empty(ci * 27 * co, device=dev)
    gx = torch.empty(b, ci, r, r, r, device=dev)
    _lib.check(lib.pvcnn_conv3d_weight_transform(P(w), co, ci, 1, P(wt), S()), 'wt')
    _lib.check(lib.pvcnn_conv3d_fwd(P(gy), P(wt), None, b, co, ci, r, P(gx), S()), 'bwd_data')
    return gx


def split_wts(w, for_bwd, ns):
    co, ci = w.shape[0], w.shape[1]
    nb = lib.pvcnn_conv3d_weight_split_bytes(co, ci, for_bwd, ns)
    wts = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.pvcnn_conv3d_weight_split(P(w), co, ci, for_bwd, ns, P(wts), S()), 'split')
    return wts


def absmax(x):
    """The amax buffer of a voxel grid (include/pvcnn_hip.h): [0] global, then one maximum per z row."""
    b, c, r = x.shape[0], x.shape[1], x.shape[2]
    out = torch.empty(lib.pvcnn_absmax_tiles_count(b, r ** 3, r), dtype=torch.int32, device=dev)
    _lib.check(lib.pvcnn_absmax_tiles(P(x), b, c, r ** 3, r, P(out), None, S()), 'absmax_tiles')
    return out


def fwd_split(x, w, bias, ns):
    b, ci, r = x.shape[0], x.shape[1], x.shape[2]
    co = w.shape[0]
    y = torch.empty(b, co, r, r, r, device=dev)
    _lib.check(lib.pvcnn_conv3d_fwd_split(P(x), P(split_wts(w, 0, ns)), P(bias), b, ci, co, r, ns, P(absmax(x)) if ns == 2 else None, r if ns == 2 else 0, P(y), None, S()), 'fwd_split')
    return y


def bwd_data_split(gy, w, ns):
    b, co, r = gy.shape[0], gy.shape[1], gy.shape[2]
    ci = w.shape[1]
    gx = torch.empty(b, ci, r, r, r, device=dev)
    _lib.check(lib.pvcnn_conv3d_fwd_split(P(gy), P(split_wts(w, 1, ns)), None, b, co, ci, r, ns, P(absmax(gy)) if ns == 2 else None, r if ns == 2 else 0, P(gx), None, S()), 'bwd_data_split')
    return gx


def wgrad_f16(x, gy, with_bias=False):
    b, ci, r = x.shape[0], x.shape[1], x.shape[2]
    co = gy.shape[1]
    gw = torch.empty(co, ci, 3, 3, 3, device=dev)
    gb = torch.empty(co, device=dev) if with_bias else None
    nb = lib.pvcnn_conv3d_bwd_weight_f16_workspace_bytes(b, ci, co, r)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    ax, ag = absmax(x), absmax(gy)      # both alive until the launch is enqueued (a freed temporary would be reused by the second)
    _lib.check(lib.pvcnn_conv3d_bwd_weight_f16(P(x), P(gy), P(ax), 0, P(ag), 0, b, ci, co, r, P(gw), P(gb), P(ws), nb, S()), 'wgrad_f16')
    return (gw, gb) if with_bias else gw


def graph_time(fn, reps=5, iters=5):
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
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    torch.manual_seed(0)
    for (b, ci, co, r) in [] if '--no-check' in sys.argv else [(2, 9, 64, 32), (1, 5, 7, 12), (2, 64, 64, 16), (1, 16, 130, 8), (1, 3, 4, 33), (1, 8, 8, 5)]:
        x = torch.randn(b, ci, r, r, r, device=dev)
        w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.1
        bias = torch.randn(co, device=dev)
        ref = F.conv3d(x.double(), w.double(), bias.double(), padding=1)
        err = (fwd(x, w, bias).double() - ref).abs().max().item() / ref.abs().max().item()
        gy = torch.randn(b, co, r, r, r, device=dev)
        xd = x.double().requires_grad_()
        F.conv3d(xd, w.double(), padding=1).backward(gy.double())
        errd = (bwd_data(gy, w).double() - xd.grad).abs().max().item() / xd.grad.abs().max().item()
        wd = w.double().requires_grad_()
        F.conv3d(x.double(), wd, padding=1).backward(gy.double())
        gw = torch.empty_like(w)
        nb = lib.pvcnn_conv3d_bwd_weight_workspace_bytes(b, ci, co, r)
        wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.pvcnn_conv3d_bwd_weight(P(x), P(gy), b, ci, co, r, P(gw), None, P(wsb), nb, S()), 'bwd_w')
        errw = (gw.double() - wd.grad).abs().max().item() / wd.grad.abs().max().item()
        print(json.dumps({'check_BCiCoR': [b, ci, co, r], 'fwd_rel_err': err, 'bwd_data_rel_err': errd, 'bwd_weight_rel_err': errw,
                          'ok': max(err, errd, errw) < 1e-5}), flush=True)
        for ns in (2, 3, 1):
            e1 = (fwd_split(x, w, bias, ns).double() - ref).abs().max().item() / ref.abs().max().item()
            e2 = (bwd_data_split(gy, w, ns).double() - xd.grad).abs().max().item() / xd.grad.abs().max().item()
            print(json.dumps({'check_split_BCiCoR': [b, ci, co, r], 'nsplit': ns, 'fwd_rel_err': e1, 'bwd_data_rel_err': e2,
                              'ok': max(e1, e2) < (1e-5 if ns != 1 else 2e-2)}), flush=True)
    for (b, ci, co, r) in [] if '--no-check' in sys.argv else [(2, 9, 64, 32), (2, 64, 64, 16), (3, 40, 70, 16), (1, 33, 130, 32), (1, 1, 1, 16)]:
        x = torch.randn(b, ci, r, r, r, device=dev)
        gy = torch.randn(b, co, r, r, r, device=dev) * 1e-3
        wd = torch.zeros(co, ci, 3, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
        F.conv3d(x.double(), wd, torch.zeros(co, device=dev, dtype=torch.float64), padding=1).backward(gy.double())
        gw, gb = wgrad_f16(x, gy, True)
        gw2, _ = wgrad_f16(x, gy, True)
        e1 = (gw.double() - wd.grad).abs().max().item() / wd.grad.abs().max().item()
        e2 = (gb.double() - gy.double().sum(dim=(0, 2, 3, 4))).abs().max().item() / gy.double().sum(dim=(0, 2, 3, 4)).abs().max().item()
        print(json.dumps({'check_wgrad_f16_BCiCoR': [b, ci, co, r], 'gw_rel_err': e1, 'gb_rel_err': e2, 'deterministic': bool(torch.equal(gw, gw2)),
                          'ok': e1 < 1e-5 and e2 < 1e-5}), flush=True)
    if '--no-check' not in sys.argv:   # f16x2 under awkward magnitudes: tiny / huge tensors, outliers, per-row weight scales, zeros
        g = torch.Generator(device=dev).manual_seed(1)
        for name, xs, wsc in [('tiny', 1e-20, 1e-6), ('huge', 1e12, 1e3), ('outlier', 1.0, 1.0), ('rows', 1.0, None), ('zero', 0.0, 1.0)]:
            x = torch.randn(2, 64, 16, 16, 16, device=dev, generator=g) * xs
            if name == 'outlier':
                x.view(-1)[12345] = 3.0e4
            w = torch.randn(64, 64, 3, 3, 3, device=dev, generator=g) * 0.05
            w = w * (wsc if wsc is not None else torch.logspace(-8, 6, 64, device=dev).view(-1, 1, 1, 1, 1))
            ref = F.conv3d(x.double(), w.double(), padding=1)
            scale = ref.abs().amax(dim=(0, 2, 3, 4), keepdim=True).clamp_min(1e-300)       # per output channel
            out = {}
            for ns in (2, 3):
                out[ns] = (((fwd_split(x, w, None, ns).double() - ref).abs() / scale).max().item())
            e0 = ((fwd(x, w, None).double() - ref).abs() / scale).max().item()
            print(json.dumps({'range_case': name, 'f16x2_err': out[2], 'bf16x3_err': out[3], 'fp32_mfma_err': e0}), flush=True)
    if '--time' in sys.argv:
        shapes = [(16, 9, 64, 32), (16, 64, 64, 32), (16, 64, 64, 16), (16, 64, 128, 16), (16, 128, 128, 16)]
        if '--shapes' in sys.argv:   # e.g. --shapes 16x64x64x16,16x128x128x16
            shapes = [tuple(int(v) for v in t.split('x')) for t in sys.argv[sys.argv.index('--shapes') + 1].split(',')]
        for (b, ci, co, r) in shapes:
            x = torch.randn(b, ci, r, r, r, device=dev)
            w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.1
            bias = torch.randn(co, device=dev)
            wt = torch.empty(ci * 27 * co, device=dev)
            y = torch.empty(b, co, r, r, r, device=dev)
            lib.pvcnn_conv3d_weight_transform(P(w), co, ci, 0, P(wt), S())
            ms = graph_time(lambda: lib.pvcnn_conv3d_fwd(P(x), P(wt), P(bias), b, ci, co, r, P(y), S()))
            fl = 2 * b * r ** 3 * 27 * ci * co
            gy = torch.randn_like(y)
            gw = torch.empty_like(w)
            nb = lib.pvcnn_conv3d_bwd_weight_workspace_bytes(b, ci, co, r)
            wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
            msw = graph_time(lambda: lib.pvcnn_conv3d_bwd_weight(P(x), P(gy), b, ci, co, r, P(gw), None, P(wsb), nb, S()))
            am = absmax(x)
            msa = graph_time(lambda: lib.pvcnn_absmax_tiles(P(x), b, ci, r ** 3, r, P(am), None, S()))
            print(json.dumps({'absmax_tiles_BCR': [b, ci, r], 'ms': round(msa, 4), 'GBps': round(x.numel() * 4 / msa / 1e6, 0)}), flush=True)
            if r in (16, 32):
                nb16 = lib.pvcnn_conv3d_bwd_weight_f16_workspace_bytes(b, ci, co, r)
                ws16 = torch.empty(nb16, dtype=torch.uint8, device=dev)
                ax, ag = absmax(x), absmax(gy)
                ms16 = graph_time(lambda: lib.pvcnn_conv3d_bwd_weight_f16(P(x), P(gy), P(ax), 0, P(ag), 0, b, ci, co, r, P(gw), None, P(ws16), nb16, S()))
                print(json.dumps({'time_wgrad_f16_BCiCoR': [b, ci, co, r], 'ms': round(ms16, 4), 'effective_TFLOPs': round(fl / ms16 / 1e9, 1),
                                  'fp32_mfma_kernel_ms': round(msw, 4), 'ws_MB': round(nb16 / 1e6, 1)}), flush=True)
            for ns, dbg in [(2, 0), (3, 0), (1, 0)]:
                wts = split_wts(w, 0, ns)
                mss = graph_time(lambda: lib.pvcnn_conv3d_fwd_split(P(x), P(wts), P(bias), b, ci, co, r, ns, P(am), r if ns == 2 else 0, P(y), None, S()))
                print(json.dumps({'time_split_BCiCoR': [b, ci, co, r], 'nsplit': ns, 'fwd_ms': round(mss, 4),
                                  'effective_TFLOPs': round(fl / mss / 1e9, 1), 'bf16_mfma_TFLOPs': round({3: 6, 2: 3, 1: 1}[ns] * fl / mss / 1e9, 1),
                                  'frac_2500TF': round({3: 6, 2: 3, 1: 1}[ns] * fl / mss / 1e9 / 2500, 3)}), flush=True)
            print(json.dumps({'time_BCiCoR': [b, ci, co, r], 'fwd_ms': round(ms, 4), 'TFLOPs': round(fl / ms / 1e9, 1),
                              'frac_157TF': round(fl / ms / 1e9 / 157.3, 3), 'bwd_weight_ms': round(msw, 4),
                              'bwd_weight_TFLOPs': round(fl / msw / 1e9, 1), 'wgrad_ws_MB': round(nb / 1e6, 1)}), flush=True)


if __name__ == '__main__':
    main()
