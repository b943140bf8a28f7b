#!/bin/bash
# Backward's weight-gradient launches on a stream of their own (functional/_sidepath.py: a parallel branch of the captured step):
# the tests that pin it bit for bit, then PVCNN_WGRAD_PATH=0|1 on one box for the four configs.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sidepath.py tests/test_gpu_graph.py tests/test_gpu_optim.py tests/test_gpu_dp_rccl.py tests/test_gpu_centers_ahead.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -2
: > $O/ab.txt
run() { # config, switch
  PVCNN_WGRAD_PATH=$2 timeout 300 python bench.py --config $1 --no-cpu-baseline --no-variants --steps 60 --warmup 10 2>$O/err_$1_$2.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wgrad path', $2, '$1', d['value'], d['ms_per_step'], d.get('eager_value'))" | tee -a $O/ab.txt
}
for i in 1 2; do for s in 0 1; do run cfg2 $s; done; done
for c in cfg3 cfg5 cfg4; do for s in 0 1; do run $c $s; done; done
