#!/bin/bash
# PVCNN++: FPS / ball query of the whole pyramid on a side stream next to the first stage's convolutions -- A/B, then the parity tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04w; mkdir -p $O
cd $R
: > $O/ab.txt
for i in 1 2; do for s in 0 1; do
  PVCNN_GEOMETRY_STREAM=$s timeout 300 python bench.py --config cfg3 --no-cpu-baseline --steps 60 --warmup 10 2>$O/err_$s.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('geometry stream', $s, 'cfg3', d['value'], d['ms_per_step'], d.get('eager_value'))" | tee -a $O/ab.txt
done; done
timeout 1200 python -m pytest tests/test_gpu_parity_as_benched.py tests/test_gpu_train_parity.py tests/test_gpu_graph.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
tail -2 $O/err_1.txt | cut -c1-300
