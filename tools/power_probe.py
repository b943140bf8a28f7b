#!/usr/bin/env python3
"""What the board reports while a kernel runs back to back: socket power and shader clock from the amdgpu hwmon / sysfs files (read-only,
sampled at ~50 Hz by a thread), next to the launch time.  Four loads of ~1.5 s each: the wide Conv3d forward kernel on random and on
constant operands, the Conv3d backward-weight kernel on random operands, a BatchNorm apply pass (HBM-bound).  The question it answers:
do the matrix kernels run at the board's power cap (then a better schedule of the same work is taken back by the clock)?
usage (needs a GPU): python tools/power_probe.py"""
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pvcnn_amd.modules.functional.backend import HipBackend  # noqa: E402

be = HipBackend()
dev = 'cuda:0'


def _first(patterns):
    for p in patterns:
        hits = sorted(glob.glob(p))
        if hits:
            return hits[0]
    return None


POWER = _first(['/sys/class/drm/card*/device/hwmon/hwmon*/power1_average', '/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'])
CAP = _first(['/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap'])
SCLK = _first(['/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input'])
DPM = _first(['/sys/class/drm/card*/device/pp_dpm_sclk'])


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def sample():
    out = {}
    v = _read(POWER) if POWER else None
    if v and v.isdigit():
        out['power_W'] = int(v) / 1e6
    v = _read(SCLK) if SCLK else None
    if v and v.isdigit():
        out['sclk_GHz'] = int(v) / 1e9
    if 'sclk_GHz' not in out and DPM:
        txt = _read(DPM) or ''
        for line in txt.splitlines():
            if line.rstrip().endswith('*'):
                try:
                    out['sclk_GHz'] = float(line.split(':')[1].split('M')[0]) / 1e3
                except (IndexError, ValueError):
                    pass
    return out


def run(name, fn, seconds=1.5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def watch():
        while not stop.is_set():
            s = sample()
            if s:
                samples.append(s)
            time.sleep(0.02)
    th = threading.Thread(target=watch)
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    e1.synchronize()
    stop.set()
    th.join()
    tail = samples[len(samples) // 3:]                       # the first third: the board is still ramping
    mean = lambda k: round(sum(s[k] for s in tail if k in s) / max(sum(1 for s in tail if k in s), 1), 3)
    row = {'load': name, 'launch_us': round(e0.elapsed_time(e1) / n * 1e3, 1), 'launches': n, 'samples': len(tail),
           'power_W_mean': mean('power_W'), 'power_W_max': round(max((s.get('power_W', 0.0) for s in tail), default=0.0), 1), 'sclk_GHz_mean': mean('sclk_GHz')}
    print(json.dumps(row), flush=True)


def main():
    cap = _read(CAP) if CAP else None
    print(json.dumps({'files': {'power': POWER, 'cap': CAP, 'sclk': SCLK, 'dpm': DPM}, 'power_cap_W': int(cap) / 1e6 if cap and cap.isdigit() else None,
                      'idle': sample()}), flush=True)
    b, c, r = 16, 64, 32
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
    wf = be._conv_wsplit(w, False, 2)
    for kind in ('randn', 'const'):
        x = torch.randn(b, c, r, r, r, device=dev) if kind == 'randn' else torch.ones(b, c, r, r, r, device=dev)
        ax = be.conv_amax(x)
        run(f'Conv3d 64->64 @ 32^3 forward (wide kernel), {kind} operands', lambda: be.conv3d_igemm_split(x, wf, None, c, 2, False, ax))
    x, gy = torch.randn(b, c, r, r, r, device=dev), torch.randn(b, c, r, r, r, device=dev)
    ax, ag = be.conv_amax(x), be.conv_amax(gy)
    run('Conv3d 64->64 @ 32^3 backward-weight, randn operands', lambda: be.conv3d_backward_weight_f16(x, gy, ax, ag))
    xp = torch.randn(16, 1472, 4096, device=dev)
    wp = be._pw_wsplit(torch.randn(512, 1472, device=dev) * 0.1, False, 2)
    ap = be.pw_amax(xp)
    run('1x1 1472->512 forward (wide kernel), randn operands', lambda: be.pwconv_gemm_split(xp, wp, None, 512, 2, False, ap))
    big = torch.randn(16, 1024, 4096, device=dev)
    out = torch.empty_like(big)
    run('copy of 268 MB (HBM-bound)', lambda: out.copy_(big))
    time.sleep(1.0)
    print(json.dumps({'idle_after': sample()}), flush=True)


if __name__ == '__main__':
    main()
