#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
 timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider -k "autocast or full_width" > $O/train_parity.log 2>&1; grep "train parity\] bf16\|passed\|failed" $O/train_parity.log | tail -4
 timeout 200 python tools/pwbench.py --shapes 16x128x1024x4096,16x1472x512x4096,16x512x256x4096 2>/dev/null | grep "^{\"f16x2" > $O/pwbench.jsonl
 timeout 200 python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32,16x64x64x16,16x64x128x16,16x128x128x16 2>$O/convcheck.err | grep "time_split\|time_wgrad" | grep -v "nsplit\": 3" > $O/convbench.jsonl
 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 60 43 > $O/bench_steady_state.txt 2>&1
ls $O
