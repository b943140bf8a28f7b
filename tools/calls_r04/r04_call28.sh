#!/bin/bash
# Frustum-PointNet's box loss + gradient as one launch (ABI v10): tests, cfg5 against the torch formulation (PVCNN_FUSED_FRUSTUM_LOSS=0)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04z; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_frustum_loss.py -x -q -p no:cacheprovider 2>&1 | tail -15 | cut -c1-220
: > $O/ab.txt
for i in 1 2; do for f in 0 1; do
  PVCNN_FUSED_FRUSTUM_LOSS=$f timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused loss', $f, 'cfg5', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
timeout 900 python -m pytest tests/test_gpu_parity_as_benched.py -x -q -p no:cacheprovider -k "cfg5 or frustum" 2>&1 | tail -3 | cut -c1-200
