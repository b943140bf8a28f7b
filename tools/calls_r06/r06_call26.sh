#!/bin/bash
# round 6, call 26: ABI v12 again: the new tests, the whole GPU suite (nothing deselected), A/B of the step with table-only amax buffers
# (default) and with word [0] everywhere (PVCNN_AMAX_GLOBAL=1), launches per step from a kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06z; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_amax_table_only.py -m gpu -q -x -p no:cacheprovider > $O/tests_new.log 2>&1; tail -3 $O/tests_new.log
for g in 0 1 0 1; do
  PVCNN_AMAX_GLOBAL=$g timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_g$g.json 2> $O/bench_g$g.err
  python - <<PY
import json; d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1]); print('amax_global=$g', d['value'], d['ms_per_step'])
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-variants > $O/bench_under_rocprof.log 2>&1
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 90 50 > $O/bench_steady_state.txt 2>&1; head -3 $O/bench_steady_state.txt
cd $R
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_all.log 2>&1; tail -5 $O/tests_all.log
