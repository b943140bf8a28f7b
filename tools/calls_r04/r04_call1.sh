#!/bin/bash
# new as-benched parity tests + RCCL-in-graph test + baseline bench lines of the round
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity_as_benched.py tests/test_gpu_dp_rccl.py tests/test_gpu_graph.py -m gpu -q -s -p no:cacheprovider > $O/new_tests.log 2>&1
grep "as benched\|passed\|failed\|Error\|assert" $O/new_tests.log | cut -c1-400 | tail -40
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_20_5.json
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_cfg5.json
python - <<'PY'
import json
for f in ('bench_20_5','bench_cfg5'):
    try:
        d=json.load(open(f'gpurun_out/r04a/{f}.json')); print(f, d['value'], d['ms_per_step'], d.get('eager_value'), d['config'].get('step_issue','')[:60])
    except Exception as e: print(f, 'ERR', e)
PY
