#!/bin/bash
# PVCNN++: how many levels of the pyramid to sample ahead (profiles/ab/r05l: an open branch costs ~3 us per main-chain launch)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_centers_ahead.py -x -q -p no:cacheprovider 2>&1 | tail -1
: > $O/ab.txt
run() { # label, env...
  l=$1; shift
  env "$@" timeout 200 python bench.py --config cfg3 --no-cpu-baseline --no-variants --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
}
for i in 1 2 3; do
  run "in line      " PVCNN_CENTERS_AHEAD=0
  run "all levels   " PVCNN_CENTERS_AHEAD=1 PVCNN_CENTERS_AHEAD_LEVELS=0
  run "first level  " PVCNN_CENTERS_AHEAD=1 PVCNN_CENTERS_AHEAD_LEVELS=1
  run "first two    " PVCNN_CENTERS_AHEAD=1 PVCNN_CENTERS_AHEAD_LEVELS=2
done
