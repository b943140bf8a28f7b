#!/bin/bash
# round 6, call 24: which torch operators launch the step's small non-pvcnn kernels (operator, shapes, innermost frame of this repository);
# the backward-weight kernel with the interleaved step at R <= 16 only: test + step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06x; mkdir -p $O
cd $R
timeout 600 python tools/step_profile.py --torch-only --rows 70 > $O/step_profile_torch_only.txt 2> $O/step_profile.err; cut -c1-330 $O/step_profile_torch_only.txt
timeout 900 python -m pytest tests/test_gpu_wgrad_pp.py tests/test_gpu_conv3d.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for pp in 1 0 1 0; do
  PVCNN_WGRAD_PP=$pp timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err
  python - <<PY
import json; d=json.loads(open('$O/bench_pp$pp.json').read().strip().splitlines()[-1]); print('pp=$pp', d['value'], d['ms_per_step'])
PY
done
