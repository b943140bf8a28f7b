#!/bin/bash
# A/B of the traversal order of the streaming passes (PVCNN_ORDER bit mask, csrc/common.h): bench value per mask, then the in-graph
# per-kernel durations of mask 0 and mask 15
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04r; mkdir -p $O
cd $R
for m in 0 15 0 15 1 2 4 8 3; do
  PVCNN_ORDER=$m timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_m$m.json
  python -c "
import json; d=json.load(open('$O/bench_m$m.json')); print('mask', $m, d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp
for m in 0 15; do
  rm -rf /tmp/kt$m; PVCNN_ORDER=$m timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$m -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
  t=$(find /tmp/kt$m -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_steady.py $t 100 60 50 > $O/steady_m$m.txt 2>&1
  python $R/tools/trace_steady.py $t 100 0 50 --by-grid bnact > $O/bnact_m$m.txt 2>&1
  head -3 $O/steady_m$m.txt
done
