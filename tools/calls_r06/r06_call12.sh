#!/bin/bash
# round 6, call 12: the adopt tests again (the Frustum one zeroes its gradients between the two steps now), and the suite's other new files
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06k; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_reference_composition.py tests/test_gpu_concat.py tests/test_gpu_fold.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -4 $O/tests.log
