#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03zg; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_concat.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for c in cfg4 cfg2 cfg3; do extra=""; [ $c != cfg2 ] && extra="--config $c --steps 40 --warmup 10"; timeout 300 python bench.py --no-cpu-baseline $extra 2>/dev/null | tail -1 > $O/bench_$c.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03zg/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
