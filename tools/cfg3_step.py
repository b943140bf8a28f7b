#!/usr/bin/env python3
"""One-off timing of BASELINE configs[2]: PVCNN++ (ball_query + grouping path) S3DIS, B=8, N=8192, fwd+bwd+Adam."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as tf
from pvcnn_amd import workload
dev = torch.device('cuda', 0)
torch.backends.cudnn.benchmark = False
torch.manual_seed(workload.SEED)
model = workload.PVCNN2(13, 6, width_multiplier=1).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
x, y = workload.make_s3dis_batch(8, 8192, device=dev, seed=workload.SEED)
def step():
    opt.zero_grad(set_to_none=True)
    loss = tf.cross_entropy(model(x), y)
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): loss = step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(json.dumps({'config': 'PVCNN2 S3DIS B=8 N=8192 fp32 fwd+bwd+Adam', 'ms_per_step': round(el / n * 1e3, 2), 'clouds_per_s': round(8 * n / el, 1), 'loss': round(float(loss), 4)}))
