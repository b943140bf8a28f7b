#!/bin/bash
# last call of the round: the whole GPU suite in one process on the final code, smoke, the driver's bench command, cfg5's lines again
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_suite_one_process.log 2>&1; grep -E "passed|failed" $O/gpu_suite_one_process.log | tail -1
timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | tee $O/smoke.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_cfg5; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_cfg5 -- python $R/bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 10 > /dev/null 2>&1
t=$(find /tmp/kt_cfg5 -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 50 --json $O/kernel_durations_cfg5.json > $O/cfg5_steady_state.txt 2>&1
(cd $R && cp $O/kernel_durations_cfg5.json profiles/kernel_durations_cfg5.json && timeout 200 python bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_cfg5.json)
head -2 $O/cfg5_steady_state.txt
python -c "
import json; d=json.load(open('$O/bench_cfg5.json')); print('cfg5', d['value'], d['ms_per_step'])"
