#!/usr/bin/env python3
"""Where do the NON-pvcnn kernels of a training step come from?  One PVCNN step under torch.profiler with shapes and Python stacks:
prints, per kernel-launching operator, device time, call count, input shapes and the innermost frame of this repository.
usage (needs a GPU): python tools/step_profile.py [--rows 40] [--config cfg2|cfg3|cfg4] [--torch-only]      """
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as tf
from torch.profiler import ProfilerActivity, profile

from pvcnn_amd import workload
from pvcnn_amd.dp import GradBucketReducer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=40)
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4'])
    ap.add_argument('--torch-only', action='store_true', help='list only the operators that are not this package\'s kernels')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.manual_seed(workload.SEED)
    if args.config == 'cfg2':
        model = workload.PVCNN(13, 6).to(dev).train()
        x, y = workload.make_s3dis_batch(16, 4096, device=dev)
    elif args.config == 'cfg3':
        model = workload.PVCNN2(13, 6).to(dev).train()
        x, y = workload.make_s3dis_batch(8, 8192, device=dev)
    else:
        model = workload.PVCNNShapeNet(50, 16, 3).to(dev).train()
        x, y = workload.make_shapenet_batch(8, 2048, device=dev)
    reducer = GradBucketReducer(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)

    def step():
        reducer.zero_grad()
        loss = tf.cross_entropy(model(x), y)
        loss.backward()
        reducer.finish()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
        dev_us = getattr(ev, 'self_device_time_total', None)
        if dev_us is None:
            dev_us = getattr(ev, 'self_cuda_time_total', 0.0)
        if dev_us <= 0:
            continue
        if args.torch_only and not ev.key.startswith('aten::'):
            continue
        frame = next((f for f in (ev.stack or []) if 'pvcnn_amd' in f or 'bench.py' in f or 'step_profile' in f), '')
        rows.append((dev_us, ev.count, ev.key, str(ev.input_shapes)[:90], frame.strip()[:110]))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f'device time of one step by operator: {total / 1e3:.3f} ms over {len(rows)} (operator, shapes, stack) groups')
    for dev_us, count, key, shapes, frame in rows[:args.rows]:
        print(f'{dev_us:9.1f} us  x{count:<3d} {key[:38]:38s} {shapes:90s} {frame}')


if __name__ == '__main__':
    main()
