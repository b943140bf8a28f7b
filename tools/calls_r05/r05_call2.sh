#!/bin/bash
# round 5, call 2: ABI v11 (folded finalize tickets, pair plans), the seam's plan memo, adopt(), per-tensor judge (flip sites / shadows),
# then the whole GPU suite and a same-box A/B of the step with / without the two launch-count changes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
export PVCNN_PARITY_DUMP=$O/parity
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fold.py tests/test_gpu_plans.py tests/test_gpu_reference_composition.py tests/test_gpu_optim.py tests/test_gpu_bnact.py tests/test_golden.py \
  -q -m gpu -p no:cacheprovider -s > $O/new_tests.log 2>&1
echo "new tests: $(grep -E 'passed|failed' $O/new_tests.log | tail -1)"
grep -E "^FAILED|^ERROR|^\[reference composition\]" $O/new_tests.log | cut -c1-600 | head -30
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1
echo "suite: $(grep -E 'passed|failed' $O/gpu_suite.log | tail -1)"
grep -E "^FAILED|^ERROR" $O/gpu_suite.log | cut -c1-300 | head -30
for v in "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  PVCNN_FOLD_FINALIZE=$1 PVCNN_PAIR_PLANS=$2 timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench_fold$1_pair$2.json
  python -c "
import json; d=json.load(open('$O/bench_fold$1_pair$2.json')); print('fold=$1 pair=$2', d['value'], d['ms_per_step'])"
done
