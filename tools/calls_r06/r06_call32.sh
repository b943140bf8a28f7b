#!/bin/bash
# round 6, call 32: the small 1x1 backward-weight launch (64 -> 64 over 65 536 points) runs 128 workgroups: its partition cap, 128 / 256 / 512
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06sm; mkdir -p $O
cd $R
for pm in 128 256 512; do PVCNN_PW_SMALL_PMAX=$pm timeout 300 python tools/pw_small_bench.py 2>>$O/err.log | tee -a $O/pw_small.jsonl; done
for pm in 128 256 128 256; do
  PVCNN_PW_SMALL_PMAX=$pm timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_$pm.json 2> $O/bench_$pm.err
  python - <<PY
import json; d=json.loads(open('$O/bench_$pm.json').read().strip().splitlines()[-1]); print('pmax=$pm', d['value'], d['ms_per_step'])
PY
done
