#!/bin/bash
# parameter gradients written into the flat buckets by the backward kernels: tests, then cfg2 / cfg3 / cfg5 against the gathering path
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04v; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_graph.py tests/test_gpu_dp_rccl.py tests/test_gpu_parity_as_benched.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
: > $O/ab.txt
for i in 1 2; do for s in 0 1; do for c in cfg2 cfg3 cfg5; do
  PVCNN_GRAD_SLOTS=$s timeout 300 python bench.py --config $c --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots', $s, '$c', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done; done
