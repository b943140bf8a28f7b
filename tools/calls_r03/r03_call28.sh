#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03z; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench100.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03z/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
