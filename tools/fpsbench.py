#!/usr/bin/env python3
"""FPS latency per dependent step ('FPS: latency (M-1 dependent steps), report us/step')."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
for (b, n, m) in [(8, 8192, 1024), (8, 1024, 256), (8, 256, 64), (8, 64, 16), (16, 4096, 1024), (64, 2048, 512), (8, 16384, 1024)]:
    c = torch.rand(b, 3, n, device='cuda:0')
    for _ in range(2): be.furthest_point_sampling(c, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): be.furthest_point_sampling(c, m)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({'BNM': [b, n, m], 'ms': round(ms, 3), 'us_per_step': round(ms * 1e3 / (m - 1), 3)}))
