#!/bin/bash
# the bench's multi-GPU code path on ONE GPU: 1-rank RCCL group, forced bucket all-reduces, whole step in one graph
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline 2>$O/plain.err | tail -1 > $O/plain.json
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --single-rank-collectives 2>$O/coll.err | tail -1 > $O/coll.json
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --single-rank-collectives --collectives-after-replay 2>$O/after.err | tail -1 > $O/after.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/torchrun.err | tail -1 > $O/torchrun.json
for f in plain coll after torchrun; do python -c "
import json
try:
    d=json.load(open('$O/$f.json')); c=d['config']; print('$f', d['value'], d['ms_per_step'], c['rccl_ranks'], c['gradient_buckets'], c['allreduce_alone_us_per_step'], c['step_issue'][:90], d.get('collective_capture_error'))
except Exception as e: print('$f ERR', e); print(open('$O/$f.err').read()[-600:])
"; done
