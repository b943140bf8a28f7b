#!/bin/bash
# pw wgrad partition-count fix: tests, op table, step + steady-state kernel table
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R; timeout 900 python -m pytest tests/test_gpu_pwconv.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -1 $O/tests.log
python tools/pwbench.py 2>/dev/null | grep "^{\"f16x2_wgrad" > $O/pwbench.jsonl
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 40 43 > $O/bench_steady_state.txt 2>&1
rm -f $O/bench_under_rocprof.log
cd $R; cut -c1-200 $O/pwbench.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03m/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
head -30 $O/bench_steady_state.txt | cut -c1-150
