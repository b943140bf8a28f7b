#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 40 43 > $O/bench_steady_state.txt 2>&1
rm -f $O/bench_under_rocprof.log
cd $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03s/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
head -24 $O/bench_steady_state.txt | cut -c1-150
