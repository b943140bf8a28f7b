#!/bin/bash
# zero-input tiles (conv fwd) + zero rows (conv wgrad): parity and effect on the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py tests/test_gpu_pwconv.py tests/test_gpu_bnact.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -5 $O/tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -x -p no:cacheprovider -k "pvconv or full_width" > $O/tests2.log 2>&1
tail -3 $O/tests2.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 45 43 --by-grid conv3d > $O/bench_steady_state.txt 2>&1
head -24 $O/bench_steady_state.txt | cut -c1-150; grep -A16 "by launch grid" $O/bench_steady_state.txt | cut -c1-170
cd $R && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_20_5.json
python -c "
import json; d=json.load(open('$O/bench_20_5.json')); print(d['value'], d['ms_per_step'])"
