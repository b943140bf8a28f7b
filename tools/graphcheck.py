#!/usr/bin/env python3
"""Eager vs captured training step: same losses, step time of each (tools; needs a GPU)."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as tf
from pvcnn_amd import workload
from pvcnn_amd.dp import GradBucketReducer
from pvcnn_amd.graph import GraphedTrainStep

dev = torch.device('cuda', 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
torch.manual_seed(0)
if cfg == 'cfg3':
    model = workload.PVCNN2(13, 6).to(dev).train(); b, n = 8, 8192
else:
    model = workload.PVCNN(13, 6).to(dev).train(); b, n = 16, 4096
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0                       # identical trajectories need identical masks
x, y = workload.make_s3dis_batch(b, n, device=dev, seed=1)
model2 = copy.deepcopy(model)

def build(mod, capturable):
    red = GradBucketReducer(mod)
    opt = torch.optim.Adam(mod.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=capturable)
    return red, opt

red, opt = build(model, False)
def eager():
    red.zero_grad(); loss = tf.cross_entropy(model(x), y); loss.backward(); red.finish(); opt.step(); return loss
le = [eager().item() for _ in range(8)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): l = eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 30
red2, opt2 = build(model2, True)
step = GraphedTrainStep(model2, lambda: tf.cross_entropy(model2(x), y), opt2, red2, warmup=3)
# the warm-up inside took 3 eager steps: losses 0..2; the capture itself does not execute; replays continue from step 3
lg = [step().item() for _ in range(5)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): l = step()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 30
print('eager losses ', [round(v, 5) for v in le])
print('graph losses ', [None] * 3 + [round(v, 5) for v in lg])
print(f'eager {te * 1e3:.3f} ms/step   graph {tg * 1e3:.3f} ms/step')
