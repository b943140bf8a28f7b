#!/bin/bash
# round 6, call 25: ABI v12 (table-only amax buffers; the backward-weight kernels take the global maxima from the tables): the new test,
# the whole GPU suite, launches per step and the step time
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06y; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_amax_table_only.py tests/test_gpu_wgrad_pp.py -m gpu -q -x -p no:cacheprovider > $O/tests_new.log 2>&1; tail -5 $O/tests_new.log
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/tests_all.log 2>&1; tail -5 $O/tests_all.log
timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'launch' in k})
PY
timeout 1500 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -x -p no:cacheprovider > $O/tests_parity.log 2>&1; tail -3 $O/tests_parity.log
