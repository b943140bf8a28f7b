#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
 timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1; grep "train parity\] bf16\|passed\|failed" $O/train_parity.log | tail -4
 timeout 200 python tools/pwbench.py --shapes 16x128x1024x4096,16x1472x512x4096,16x512x256x4096 2>/dev/null | grep "^{\"f16x2" > $O/pwbench.jsonl
 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json
 for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 10 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 90 43 > $O/bench_steady_state.txt 2>&1
rm -rf /tmp/kt5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python $R/bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 10 > $O/cfg5_under_rocprof.log 2>&1
t=$(find /tmp/kt5 -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 43 > $O/cfg5_steady_state.txt 2>&1
ls $O
