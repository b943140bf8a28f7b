#!/bin/bash
# A/B of the BatchNorm backward traversal orders (PVCNN_ORDER: bits 3-4 reduce, bits 5-6 backward apply, bit 0 / 7 forward apply)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04s; mkdir -p $O
cd $R
for m in 0 8 40 72 16 48 80 0 8 73 200 72; do
  PVCNN_ORDER=$m timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_m$m.json
  python -c "
import json; d=json.load(open('$O/bench_m$m.json')); print('mask', $m, d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
