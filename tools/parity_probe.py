"""Per-tensor train-mode parity table (HIP vs fp32 oracle stack vs fp64 truth) in network order.
Usage on the GPU box: python tools/parity_probe.py PVCNN|PVCNN2|PVCNNShapeNet [--no-fuse]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as tf
import test_gpu_train_parity as t
from oracle import oracle_backend
from pvcnn_amd import workload

name = sys.argv[1] if len(sys.argv) > 1 else 'PVCNN'
orc = oracle_backend.OracleBackend()
from pvcnn_amd.modules.functional import backend as seam
for flag in sys.argv[2:]:
    if flag.startswith('--off='):
        for attr in flag[6:].split(','):
            setattr(seam._backend, attr, False)
            print('disabled', attr)
build, batch = t.NETS[name]
x0, y0 = batch(workload)
def make(dev, dtype):
    x = x0.clone().to(dev, dtype).requires_grad_()
    return x, x, y0.to(dev)
(lg, gg), (lc, gc), (lt, gt) = t._run_three(lambda: build(workload), make, tf.cross_entropy, orc, pin_winners=True)
print('loss', lg, lc, lt)
for k in gt:
    sc = t._scale(gt, k)
    a = (gg[k] - gt[k]).abs().max().item() / sc
    c = (gc[k] - gt[k]).abs().max().item() / sc
    print(f'{a:9.2e} {c:9.2e}  {"**" if a > 10 * c and a > 1e-5 else "  "} {k}')
