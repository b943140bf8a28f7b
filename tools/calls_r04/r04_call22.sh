#!/bin/bash
# experiment: PVConv's point branch on a side stream (parallel paths in the captured step)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04u; mkdir -p $O
cd $R
for f in 0 1 0 1; do
  for c in cfg2 cfg3; do
    PVCNN_FORK_POINT_BRANCH=$f timeout 300 python bench.py --config $c --no-cpu-baseline --steps 60 --warmup 10 2>$O/err_${c}_$f.txt | tail -1 > $O/b_${c}_$f.json
    python -c "
import json; d=json.load(open('$O/b_${c}_$f.json')); print('fork', $f, '$c', d['value'], d['ms_per_step'])" 2>&1 | tail -1 | tee -a $O/ab.txt
  done
done
PVCNN_FORK_POINT_BRANCH=1 timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_parity_as_benched.py -x -q -p no:cacheprovider 2>&1 | tail -3
tail -3 $O/err_cfg2_1.txt
