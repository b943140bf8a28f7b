#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -2
 timeout 1200 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1; grep "passed\|failed" $O/train_parity.log | tail -2
 for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_$c.json; done
 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json)
rm -rf /tmp/kt5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python $R/bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 10 > /dev/null 2>&1
t=$(find /tmp/kt5 -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 40 43 > $O/cfg5_steady_state.txt 2>&1
cd $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03x/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
head -16 $O/cfg5_steady_state.txt | cut -c1-150
