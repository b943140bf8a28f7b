#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
 timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider -k "full_width or autocast or pvconv" > $O/train_parity.log 2>&1; grep "passed\|failed" $O/train_parity.log | tail -2
 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 60 43 > $O/bench_steady_state.txt 2>&1
rm -f $O/bench_under_rocprof.log; ls $O
