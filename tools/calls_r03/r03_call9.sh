#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
 timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1; grep "passed\|failed" $O/train_parity.log | tail -2
 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
for c in cfg3 cfg4; do
  rm -rf /tmp/kt_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- python $R/bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 > $O/${c}_under_rocprof.log 2>&1
  t=$(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 43 > $O/${c}_steady_state.txt 2>&1
  (cd $R && timeout 300 python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json)
done
rm -f $O/*_under_rocprof.log; ls $O
