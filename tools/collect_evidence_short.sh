#!/bin/bash
# Short evidence refresh (one gpurun call, ~90 s): the bench line, rocprofv3 kernel stats + steady-state reduction of the same
# command, and (AB=1) the same bench with the round's two opt-in fusions switched on (same box: an A/B that box-to-box clock
# spread cannot blur).  Outputs -> gpurun_out/r02s/.  The PMC passes / op tables of tools/collect_evidence.sh are not repeated.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
if [ -n "$AB" ]; then   # A/B: the two opt-in fusions switched on (same box, back to back)
  (cd $R && PVCNN_FOLD_BN=1 PVCNN_BWD_AMAX=1 timeout 100 $BENCH --no-cpu-baseline 2>$O/bench_fold.err | tail -1 > $O/bench_fold.json)
fi
(cd $R && timeout 200 $BENCH 2>$O/bench.err | tail -1 > $O/bench.json)
rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $BENCH --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -80 $f > $O/bench_kernel_stats.csv
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 70 20 > $O/bench_steady_state.txt
rm -f $O/bench_under_rocprof.log
python - <<PY
import json
for n in ('bench_fold', 'bench', 'bench_under_rocprof'):
    try:
        d = json.load(open('$O/' + n + '.json')); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma']['frac'])
    except Exception as e:
        print(n, 'unreadable', e)
PY
head -12 $O/bench_steady_state.txt
