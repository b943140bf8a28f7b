#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py tests/test_gpu_pwconv.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
 timeout 600 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider -k "autocast" > $O/train_parity.log 2>&1; grep "train parity" $O/train_parity.log | tail -3; tail -3 $O/train_parity.log
 for pf in 1 2 3; do PVCNN_PW_PF=$pf timeout 200 python tools/pwbench.py --shapes 16x128x1024x4096,16x1472x512x4096,16x512x256x4096 2>/dev/null | grep "^{\"f16x2" > $O/pwbench_pf$pf.jsonl; done
 for pf in 1 2 3; do PVCNN_PW_PF=$pf timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_pf$pf.json; done
 timeout 200 python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32,16x64x64x16,16x64x128x16,16x128x128x16 2>$O/convcheck.err | grep "time_\|absmax" > $O/convbench.jsonl
 for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 10 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 80 43 > $O/bench_steady_state.txt 2>&1
ls $O
