#!/bin/bash
# round 5, call 7: the last point stage written into its slice of the classifier's concatenation: tests, same-box A/B, in-graph times
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_concat.py tests/test_gpu_bnact.py tests/test_gpu_reference_composition.py tests/test_gpu_models.py tests/test_gpu_graph.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "tests: $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" $O/tests.log | head -12 | cut -c1-300
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench_$name.json
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
run slot_on A=0
run slot_off PVCNN_CONCAT_SLOT=0
run slot_on2 A=0
for c in cfg4; do
  env A=0 timeout 300 python bench.py --config $c --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_${c}_on.json
  env PVCNN_CONCAT_SLOT=0 timeout 300 python bench.py --config $c --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_${c}_off.json
  python -c "
import json
for n in ('on','off'):
    d=json.load(open('$O/bench_${c}_%s.json'%n)); print('$c slot', n, d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-variants --steps 40 --warmup 10 > /dev/null 2>&1
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steady.py $t 40 70 20 --by-grid concat_points > $O/trace.txt 2>&1
grep -E "concat_points|last 40|bnact_apply_pb_kernel<false, false, true>" $O/trace.txt | cut -c1-150
