"""Do two independent kernels on two streams overlap -- eagerly, and as parallel paths of a captured hipGraph?
Probe: furthest point sampling of ONE cloud (one workgroup, ~0.9 ms, 255 CUs idle) twice.  python tools/graph_concurrency_probe.py"""
import sys
import time

import torch

sys.path.insert(0, '.')
from pvcnn_amd.modules.functional.backend import HipBackend  # noqa: E402

be = HipBackend()
dev = 'cuda:0'
a = torch.rand(1, 3, 8192, device=dev)
b = torch.rand(1, 3, 8192, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def serial():
    be.furthest_point_sampling(a, 1024)
    be.furthest_point_sampling(b, 1024)


def forked():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        x = be.furthest_point_sampling(a, 1024)
    y = be.furthest_point_sampling(b, 1024)
    cur.wait_stream(s1)
    return x, y


print('one launch          %.3f ms' % timed(lambda: be.furthest_point_sampling(a, 1024)))
print('eager, one stream   %.3f ms' % timed(serial))
print('eager, two streams  %.3f ms' % timed(forked))
for name, f in (('graph, one stream ', serial), ('graph, two streams', forked)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s2):
            f()
    print('%s  %.3f ms' % (name, timed(g.replay)))
