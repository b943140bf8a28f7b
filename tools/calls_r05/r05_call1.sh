#!/bin/bash
# round 5, call 1: the per-tensor network bars, the reference composition (test + bench variants), the tightened golden test; baseline bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
export PVCNN_PARITY_DUMP=$O/parity
timeout 1500 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_parity_as_benched.py tests/test_gpu_reference_composition.py tests/test_golden.py \
  -q -m gpu -p no:cacheprovider -s > $O/parity_tests.log 2>&1
grep -E "passed|failed" $O/parity_tests.log | tail -1
grep -E "^\[per tensor\]|^\[reference composition\]|^FAILED|Error" $O/parity_tests.log | cut -c1-400 | head -60
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json
python -c "
import json
d=json.load(open('$O/bench_20_5.json')); print('bench', d['value'], d['ms_per_step'], 'refcomp', d.get('reference_composition'), 'fp32', d.get('fp32_mfma'), 'roofline', d['roofline']['frac'], d['roofline']['priced_on'], d['roofline']['in_graph_trace'])"
