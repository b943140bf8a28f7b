#!/bin/bash
# A/B of the pipelined wide f16x2 1x1 GEMM (PVCNN_PW_PIPE=1, default) against pw_gemm_bf16_kernel<2,4,2,VEC> (=0)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_range.py tests/test_gpu_bnact.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for p in 1 0; do
  PVCNN_PW_PIPE=$p python tools/pwbench.py 2>/dev/null | grep "^{\"f16x2_BCiCoN" > $O/pwbench_pipe$p.jsonl
  PVCNN_PW_PIPE=$p timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pipe$p.json
  PVCNN_PW_PIPE=$p timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench100_pipe$p.json
done
timeout 600 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider -k "full_width" > $O/train_parity.log 2>&1; grep "passed\|failed" $O/train_parity.log | tail -1
cut -c1-330 $O/pwbench_pipe*.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03k/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
PY
