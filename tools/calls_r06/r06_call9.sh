#!/bin/bash
# round 6, call 9: the wide Conv3d kernel: bit-identity test again, and what a chunk spends on what (ablation builds)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_wide.py -m gpu -q -x -p no:cacheprovider > $O/tests_wide.log 2>&1; tail -3 $O/tests_wide.log
for ab in 0 1 2 4 8 16 31 0; do
  PVCNN_CONV_ABLATE=$ab timeout 200 python tools/phase_probe.py conv 2>/dev/null | tee -a $O/ablate_conv_fwd.jsonl
done
