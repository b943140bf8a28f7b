#!/bin/bash
# A/B: concat_points walking one cloud at a time (channel groups innermost)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04y; mkdir -p $O; cd $R; : > $O/ab.txt
for i in 1 2 3; do for m in 0 1; do
  PVCNN_CONCAT_ORDER=$m timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('concat order', $m, d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
