#!/bin/bash
# round 6, call 30: the backward-weight kernel with its requests three steps ahead (four register sets): tests, per-layer time, step A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06ah; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wgrad_pp.py tests/test_gpu_amax_table_only.py tests/test_gpu_conv3d.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for pp in 0 1; do PVCNN_WGRAD_PP=$pp timeout 300 python tools/wgrad_only.py --shapes 16x9x64x32,16x9x64x16,16x64x64x16,16x64x128x16,16x128x128x16,16x64x64x32 2>/dev/null | sed "s/^{/{\"PVCNN_WGRAD_PP\": $pp, /"; done | tee $O/wgrad_ahead.jsonl
for pp in 1 0 1 0; do
  PVCNN_WGRAD_PP=$pp timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err
  python - <<PY
import json; d=json.loads(open('$O/bench_pp$pp.json').read().strip().splitlines()[-1]); print('pp=$pp', d['value'], d['ms_per_step'])
PY
done
