#!/usr/bin/env python3
"""Where the host time of an eagerly issued PVCNN step goes (cProfile of a few eager steps, optionally after a graph capture).
usage: eager_profile.py [--after-graph]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as tf
from pvcnn_amd import workload
from pvcnn_amd.dp import GradBucketReducer
from pvcnn_amd.optim import FlatAdam
from pvcnn_amd.graph import GraphedTrainStep
dev = 'cuda:0'
torch.manual_seed(0)
model = workload.PVCNN(13, 6, 1).to(dev).train()
x, y = workload.make_s3dis_batch(16, 4096, device=dev)
red = GradBucketReducer(model)
opt = FlatAdam(red, lr=1e-3, weight_decay=1e-5)


def eager():
    red.zero_grad()
    loss = tf.cross_entropy(model(x), y)
    loss.backward()
    red.finish()
    opt.step()


for _ in range(3):
    eager()
torch.cuda.synchronize()
if '--after-graph' in sys.argv:
    g = GraphedTrainStep(model, lambda: tf.cross_entropy(model(x), y), opt, red, warmup=3)
    for _ in range(5):
        g()
    torch.cuda.synchronize()
for _ in range(3):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eager()
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eager()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
