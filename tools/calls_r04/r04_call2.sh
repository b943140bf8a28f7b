#!/bin/bash
# rolling-commit R=32 gather + spill-free R=16 rows gather: parity, op bench, in-graph times; new optimizer tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gather_pipe.py tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_bnact.py tests/test_gpu_optim.py tests/test_gpu_fuzz.py tests/test_gpu_plans.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -5 $O/tests.log | cut -c1-300
python tools/opbench.py --ops devox_fwd,devox_fwd_eval 2>/dev/null | grep median > $O/opbench_devox.jsonl
python tools/opbench.py --ops devox_fwd --shapes 8x64x2048x32,16x128x4096x32,64x64x2048x32 2>/dev/null | grep median >> $O/opbench_devox.jsonl
cut -c1-200 $O/opbench_devox.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 60 43 --by-grid gather_lds > $O/bench_steady_state.txt 2>&1
head -3 $O/bench_steady_state.txt; grep -A12 "by launch grid" $O/bench_steady_state.txt
cd $R && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_20_5.json
python -c "
import json; d=json.load(open('$O/bench_20_5.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'])"
