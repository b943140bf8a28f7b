#!/bin/bash
# final check on the last commit: the whole GPU suite in one process (the driver's command), smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/suite.log 2>&1; tail -2 $O/suite.log | cut -c1-200
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20_5.json
python -c "
import json; d=json.load(open('$O/bench_20_5.json')); print(d['value'], d['ms_per_step'], d['eager_value'], d['roofline']['frac'], d['roofline_mfma']['frac'], d['cpu_baseline']['value'])"
