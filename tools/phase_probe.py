#!/usr/bin/env python3
"""Where the cycles of a kernel go: phase clocks.

tools/probe/ builds ONE translation unit of pvcnn_amd/csrc with -DPVCNN_PHASE_PROBE: the PVCNN_PROBE(k) stamps in the kernel source
(nothing in the product build) become s_memtime reads, every wave adds the time between consecutive stamps to slot k and, at its end,
its slots to a 32-word device buffer.  This tool prepares the operands with the PRODUCT library (weight images, amax buffers: same
formats), launches the probe build's entry point on them and prints the shares, next to the same launch timed un-instrumented.

    python tools/phase_probe.py pw [--shape BxCIxCOxN] [--bwd-data]     the 1x1 GEMM (pointwise_bf16.hip)
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pvcnn_amd.modules.functional.backend import HipBackend  # noqa: E402

PW_SLOTS = {3: "a step's 48 MFMAs with its requests, conversion and stores between them", 4: 'the step barrier', 5: "the item's epilogue", 6: 'prologue'}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _time(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def probe_pw(shape, bwd_data):
    be = HipBackend()
    ablate = int(os.environ.get('PVCNN_PW_ABLATE', '0'))
    clocks = '--ablate' not in sys.argv
    lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', ('libprobe' if clocks else 'libablate') + '_pointwise_bf16.so'))
    b, ci, co, n = shape
    dev = 'cuda:0'
    g = torch.Generator(device=dev).manual_seed(3)
    w = torch.randn(co, ci, device=dev, generator=g) * 0.1
    x = torch.randn(b, co if bwd_data else ci, n, device=dev, generator=g)
    k, m = (co, ci) if bwd_data else (ci, co)
    bias = None if bwd_data else torch.randn(co, device=dev, generator=g)
    wts = be._pw_wsplit(w, bwd_data, 2)
    amax = be.pw_amax(x)
    y = torch.empty(b, m, n, device=dev)
    buf = torch.zeros(32, dtype=torch.int64, device=dev)
    assert not clocks or lib.pvcnn_probe_set_buffer(_p(buf)) == 0
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        rc = lib.pvcnn_pwconv_fwd_split(_p(x), _p(wts), _p(bias) if bias is not None else None, b, k, m, n, 2, _p(amax), 256, _p(y), None, stream)
        assert rc == 0, rc
    launch()
    torch.cuda.synchronize()
    ref = be.pwconv_gemm_split(x, wts, bias, m, 2, False, amax)
    assert ablate or torch.equal(ref, y), 'the probe build computes something else'
    buf.zero_()
    launch()
    torch.cuda.synchronize()
    slots = buf.cpu().tolist()
    probed_us = _time(launch)
    plain_us = _time(lambda: be.pwconv_gemm_split(x, wts, bias, m, 2, False, amax))
    if not clocks:
        print(json.dumps({'ablate': ablate, 'BCiCoN': list(shape), 'direction': 'backward-data' if bwd_data else 'forward',
                          'launch_us': round(_time(launch, 20), 1), 'launch_us_product': round(_time(lambda: be.pwconv_gemm_split(x, wts, bias, m, 2, False, amax), 20), 1)}))
        return
    waves, total = slots[31], sum(slots[:16])
    rows = [{'slot': k_, 'what': PW_SLOTS.get(k_, ''), 'share': round(v / total, 4), 'cycles_per_wave': round(v / max(waves, 1))}
            for k_, v in enumerate(slots[:16]) if v]
    steps_per_wave = None
    if k % 64 == 0 and m >= 256:
        items = b * (n // 256) * ((m + 255) // 256)
        steps_per_wave = items * (k // 16) / (waves / 4)
    out = {'ablate': ablate, 'kernel': '1x1 GEMM f16x2 (pointwise_bf16.hip)', 'BCiCoN': list(shape), 'direction': 'backward-data' if bwd_data else 'forward', 'K': k, 'M': m,
           'waves': waves, 'cycles_per_wave': round(total / max(waves, 1)), 'steps_per_wave': steps_per_wave,
           'cycles_per_step': round(total / max(waves, 1) / steps_per_wave) if steps_per_wave else None,
           'ideal_mfma_cycles_per_step': 48 * 32, 'launch_us_instrumented': round(probed_us, 1), 'launch_us_product': round(plain_us, 1), 'phases': rows}
    print(json.dumps(out))
    for r in rows:
        print(f"    slot {r['slot']:2d}  {100 * r['share']:5.1f} %  {r['cycles_per_wave']:>10d} cycles/wave   {r['what']}", file=sys.stderr)


def ablate_conv(shape, bwd_data):
    """The R = 32 Conv3d launch from the ablation build (tools/probe/libablate_conv3d_bf16.so, PVCNN_CONV_ABLATE=<bits>: 1 no A requests,
    2 no row requests, 4 no conversion / tile store, 8 no B reads, 16 no chunk barrier, 31 MFMAs only) next to the product's."""
    be = HipBackend()
    lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libablate_conv3d_bf16.so'))
    b, ci, co, r = shape
    dev = 'cuda:0'
    g = torch.Generator(device=dev).manual_seed(3)
    w = torch.randn(co, ci, 3, 3, 3, device=dev, generator=g) * 0.05
    x = torch.randn(b, co if bwd_data else ci, r, r, r, device=dev, generator=g)
    k, m = (co, ci) if bwd_data else (ci, co)
    bias = None if bwd_data else torch.randn(co, device=dev, generator=g)
    wts = be._conv_wsplit(w, bwd_data, 2)
    amax = be.conv_amax(x)
    y = torch.empty(b, m, r, r, r, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ablate = int(os.environ.get('PVCNN_CONV_ABLATE', '0'))

    def launch():
        rc = lib.pvcnn_conv3d_fwd_split(_p(x), _p(wts), _p(bias) if bias is not None else None, b, k, m, r, 2, _p(amax), r, _p(y), None, stream)
        assert rc == 0, rc
    launch()
    torch.cuda.synchronize()
    ref = be.conv3d_igemm_split(x, wts, bias, m, 2, False, amax)
    assert ablate or torch.equal(ref, y), 'the ablation build with nothing left out computes something else'
    print(json.dumps({'ablate': ablate, 'BCiCoR': list(shape), 'direction': 'backward-data' if bwd_data else 'forward',
                      'launch_us': round(_time(launch, 20), 1), 'launch_us_product': round(_time(lambda: be.conv3d_igemm_split(x, wts, bias, m, 2, False, amax), 20), 1)}))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'pw'
    shape = (16, 1472, 512, 4096)
    if '--shape' in sys.argv:
        shape = tuple(int(v) for v in sys.argv[sys.argv.index('--shape') + 1].split('x'))
    if what == 'pw':
        probe_pw(shape, '--bwd-data' in sys.argv)
    elif what == 'conv':
        ablate_conv(shape if '--shape' in sys.argv else (16, 64, 64, 32), '--bwd-data' in sys.argv)
    else:
        raise SystemExit(__doc__)
