#!/bin/bash
# round 5, call 8: cfg5's segmentation net with row maxima + concatenation slot: every Frustum test, same-box A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity_as_benched.py tests/test_gpu_train_parity.py tests/test_gpu_models.py tests/test_gpu_frustum_loss.py tests/test_gpu_logits_mask.py tests/test_gpu_graph.py \
  -q -m gpu -p no:cacheprovider -s -k "frustum or Frustum or cfg5 or graph" > $O/tests.log 2>&1
echo "tests: $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^FAILED|^ERROR|^\[per tensor\]" $O/tests.log | cut -c1-330 | head -12
for v in 1 0 1; do
  PVCNN_CONCAT_SLOT=$v timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_cfg5_slot$v.json
  python -c "
import json; d=json.load(open('$O/bench_cfg5_slot$v.json')); print('cfg5 slot=$v', d['value'], d['ms_per_step'])"
done
