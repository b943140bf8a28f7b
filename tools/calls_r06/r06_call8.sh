#!/bin/bash
# round 6, call 8: the persistent R = 32 Conv3d kernel: bit-identity, the conv tests, its speed next to the two-workgroup kernel, the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_wide.py -m gpu -q -x -p no:cacheprovider > $O/tests_wide.log 2>&1; tail -12 $O/tests_wide.log
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -4 $O/tests.log
for w in 1 0; do PVCNN_CONV_WIDE=$w timeout 300 python tools/conv_only.py 16x64x64x32 2>/dev/null | tee -a $O/conv_only.jsonl; done
for w in 1 0; do PVCNN_CONV_WIDE=$w timeout 300 python tools/conv_only.py 16x64x64x32 --relu 2>/dev/null | tee -a $O/conv_only.jsonl; done
for w in 1 0 1 0; do
  PVCNN_CONV_WIDE=$w timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  python -c "import json;d=json.loads(open('$O/bench_wide$w.json').read().strip().splitlines()[-1]);print('conv wide=$w',d['value'],d['ms_per_step'])"
done
