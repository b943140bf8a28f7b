#!/bin/bash
# the other BASELINE configs once more, with their in-graph launch durations written for bench.py (kernel_durations_cfgN.json)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
for c in cfg3 cfg4 cfg5; do
  rm -rf /tmp/kt_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- $BENCH --config $c --no-cpu-baseline --steps 40 --warmup 10 > $O/${c}_under_rocprof.log 2>&1
  t=$(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 50 --json $O/kernel_durations_$c.json > $O/${c}_steady_state.txt 2>&1
  cp $O/kernel_durations_$c.json $R/profiles/kernel_durations_$c.json
  (cd $R && timeout 300 python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_$c.json)
  python -c "
import json; d=json.load(open('$O/bench_$c.json')); r=d['roofline']; print('$c', d['value'], d['ms_per_step'], r['shape_BCNR'], r['frac'], r['priced_on_us'], r['avg_us'], r['in_graph_us'])"
done
rm -f $O/*_under_rocprof.log
