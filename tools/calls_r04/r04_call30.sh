#!/bin/bash
# plain-bf16 weight images through the weight bank (one launch per kind and step): tests, cfg5 A/B, cfg2 sanity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04aa; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_optim.py tests/test_gpu_parity_as_benched.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -1
: > $O/ab.txt
for i in 1 2; do
  timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
