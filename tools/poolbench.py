import sys, torch
sys.path.insert(0, '.')
from pvcnn_amd.modules.functional.backend import HipBackend
be = HipBackend()
x = torch.relu(torch.randn(16, 1024, 4096, device='cuda:0'))
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('row_argmax us', round(t(lambda: be.row_argmax(x)), 1), ' torch.max us', round(t(lambda: x.max(dim=-1)), 1))
w = be.row_argmax(x); assert torch.equal(w, x.max(dim=-1).indices)
y = torch.relu(torch.randn(8, 64, 1024, 32, device='cuda:0'))
print('neighbor_max us', round(t(lambda: be.neighbor_max_forward(y)), 1), ' torch.max us', round(t(lambda: y.max(dim=-1)), 1))
