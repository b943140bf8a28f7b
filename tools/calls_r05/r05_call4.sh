#!/bin/bash
# round 5, call 4: the fold with ONE wave doing publish / ticket / combine, fixed tests; same-box A/B of the launch-count changes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
export PVCNN_PARITY_DUMP=$O/parity
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fold.py tests/test_gpu_plans.py tests/test_gpu_reference_composition.py tests/test_gpu_bnact.py \
  -q -m gpu -p no:cacheprovider -s > $O/new_tests.log 2>&1
echo "new tests: $(grep -E 'passed|failed' $O/new_tests.log | tail -1)"
grep -E "^FAILED|^ERROR|reference composition\] cfg" $O/new_tests.log | cut -c1-500 | head -30
for v in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $v
  PVCNN_FOLD_FINALIZE=$1 PVCNN_PAIR_PLANS=$2 timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench_fold$1_pair$2.json
  python -c "
import json; d=json.load(open('$O/bench_fold$1_pair$2.json')); print('fold=$1 pair=$2', d['value'], d['ms_per_step'])"
done
for c in cfg3 cfg5; do
  for f in 1 0; do
    PVCNN_FOLD_FINALIZE=$f timeout 300 python bench.py --config $c --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_${c}_fold$f.json
    python -c "
import json; d=json.load(open('$O/bench_${c}_fold$f.json')); print('$c fold=$f', d['value'], d['ms_per_step'])"
  done
done
