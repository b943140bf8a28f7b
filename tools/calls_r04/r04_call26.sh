#!/bin/bash
# after the last host-side change (slots released by rearm): optimizer / graph / data-parallel / as-benched tests, the driver's bench command
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04x; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_graph.py tests/test_gpu_dp_rccl.py tests/test_gpu_parity_as_benched.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20_5.json
python -c "
import json; d=json.load(open('$O/bench_20_5.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma']['frac'])"
