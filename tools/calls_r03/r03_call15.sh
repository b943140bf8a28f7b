#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03r; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_pwconv.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -1 $O/tests.log
python tools/pwbench.py 2>/dev/null | grep "^{\"f16x2_wgrad" > $O/pwbench.jsonl
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench100.json
cut -c1-200 $O/pwbench.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03r/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
