#!/bin/bash
# PVCNN / ShapeNet PVCNN: grid coordinates + the two scatter plans of the later resolutions ahead on a stream of their own
# (pvcnn_amd.workload.plans_ahead) -- tests, then PVCNN_PLANS_AHEAD=0|1 on one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_centers_ahead.py tests/test_gpu_graph.py tests/test_gpu_plans.py tests/test_gpu_voxel_coords.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -2
: > $O/ab.txt
run() { # config, switch
  PVCNN_PLANS_AHEAD=$2 timeout 300 python bench.py --config $1 --no-cpu-baseline --no-variants --steps 100 --warmup 20 2>$O/err_$1_$2.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plans ahead', $2, '$1', d['value'], d['ms_per_step'], d.get('eager_value'))" | tee -a $O/ab.txt
}
for i in 1 2 3; do for s in 0 1; do run cfg2 $s; done; done
for i in 1 2; do for s in 0 1; do run cfg4 $s; done; done
