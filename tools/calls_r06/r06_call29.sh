#!/bin/bash
# round 6, call 29: the packed first-layer backward-weight kernel without the staging of channel rows that do not exist: test, time
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06pk; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wgrad_pp.py tests/test_gpu_amax_table_only.py tests/test_gpu_conv3d.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for pp in 0 1; do PVCNN_WGRAD_PP=$pp timeout 300 python tools/wgrad_only.py --shapes 16x9x64x32,16x9x64x16,16x64x64x16 2>/dev/null | sed "s/^{/{\"PVCNN_WGRAD_PP\": $pp, /"; done | tee $O/wgrad_pack.jsonl
