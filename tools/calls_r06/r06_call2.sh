#!/bin/bash
# round 6, call 2: the wide persistent 1x1 GEMM -- bit-identity, the GEMM tests, its speed next to the 128-row kernel, the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pw_wide.py tests/test_gpu_pwconv.py tests/test_gpu_range.py tests/test_gpu_reference_composition.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -15 $O/tests.log
for w in 1 0; do
  PVCNN_PW_WIDE=$w timeout 600 python tools/pwbench.py --shapes 16x1472x512x4096,16x128x1024x4096,16x512x256x4096 2>&1 | grep '"f16x2_BCiCoN"' > $O/pwbench_wide$w.jsonl
  cat $O/pwbench_wide$w.jsonl | cut -c1-260
done
for w in 1 0; do
  PVCNN_PW_WIDE=$w timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  python -c "import json;d=json.loads(open('$O/bench_wide$w.json').read().strip().splitlines()[-1]);print('wide=$w',d['value'],d['ms_per_step'])"
done
