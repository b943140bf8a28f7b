#!/bin/bash
# round 5, call 6: dense head with 16-byte staging and eight weight loads in flight: tests, A/B against the modules, kernel times
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_reference_composition.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "tests: $(grep -E 'passed|failed' $O/tests.log | tail -1)"
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench_$name.json
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
run dense_on A=0
run dense_off PVCNN_DENSE_HEAD=0
run dense_on2 A=0
env A=0 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_cfg5_on.json
env PVCNN_DENSE_HEAD=0 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_cfg5_off.json
python -c "
import json
for n in ('on','off'):
    d=json.load(open('$O/bench_cfg5_%s.json'%n)); print('cfg5 dense', n, d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-variants --steps 40 --warmup 10 > /dev/null 2>&1
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steady.py $t 40 70 20 --by-grid dense_bn > $O/trace.txt 2>&1
grep -E "dense_bn|last 40" $O/trace.txt | cut -c1-150
