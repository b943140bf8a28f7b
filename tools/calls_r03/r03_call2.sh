#!/bin/bash
# Round 3, second GPU call: the whole GPU test-suite on the amax-buffer code (ABI v5), then bench + steady state.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -30 $O/gpu_tests.log)
(cd $R && timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1; tail -5 $O/train_parity.log)
(cd $R && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 80 43 > $O/bench_steady_state.txt 2>&1
(cd $R && timeout 200 python tools/convcheck.py --time --no-check --shapes 16x64x64x32,16x64x64x16,16x128x128x16 2>/dev/null | grep "time_\|absmax" > $O/convbench.jsonl
 for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 10 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done)
ls $O
