#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03u; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py tests/test_gpu_bnact.py -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
python tools/convcheck.py --time --no-check --shapes 16x64x64x16,16x64x128x16,16x128x64x16,16x128x128x16,32x64x64x12,32x64x128x12 2>/dev/null | grep "time_split.*nsplit\": 2" > $O/convbench.jsonl
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench100.json
cut -c1-200 $O/convbench.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03u/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
