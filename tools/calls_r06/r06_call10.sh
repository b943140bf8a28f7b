#!/bin/bash
# round 6, call 10: the wide Conv3d kernel also at R = 16: tests, speed per layer shape, the step with and without
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_wide.py -m gpu -q -x -p no:cacheprovider > $O/tests_wide.log 2>&1; tail -8 $O/tests_wide.log
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py tests/test_gpu_bnact.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -4 $O/tests.log
for sh in 16x64x64x16 16x64x128x16 16x128x128x16; do
  for w in 1 0; do PVCNN_CONV_WIDE16=$w timeout 300 python tools/conv_only.py $sh 2>/dev/null | tee -a $O/conv_only.jsonl; done
done
for w in 1 0 1 0; do
  PVCNN_CONV_WIDE16=$w timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  python -c "import json;d=json.loads(open('$O/bench_wide$w.json').read().strip().splitlines()[-1]);print('conv wide16=$w',d['value'],d['ms_per_step'])"
done
