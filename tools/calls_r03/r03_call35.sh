#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03zf; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_range.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -p no:cacheprovider -k "autocast" 2>&1 | tail -2
for c in cfg5 cfg2; do extra=""; [ $c != cfg2 ] && extra="--config $c --steps 40 --warmup 10"; timeout 300 python bench.py --no-cpu-baseline $extra 2>/dev/null | tail -1 > $O/bench_$c.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03zf/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
