#!/bin/bash
# final state: the whole GPU suite in one process (the driver's command), smoke, then the bench lines / traces of the evidence set again
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_suite_one_process.log 2>&1; grep -E "passed|failed" $O/gpu_suite_one_process.log | tail -1
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | tee $O/smoke.log
ROUND=r04 QUICK=1 bash $R/tools/collect_evidence.sh > $O/collect_quick.log 2>&1
python -c "
import json
for f in ('bench_20_5','bench','bench_cfg3','bench_cfg4','bench_cfg5'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'])"
head -2 $O/bench_steady_state.txt
