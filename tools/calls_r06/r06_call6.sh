#!/bin/bash
# round 6, call 6: what a step of the wide 1x1 GEMM spends on what -- ablation builds (no clocks), 1472 -> 512 forward
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06f; mkdir -p $O
cd $R
for ab in 0 1 2 4 6 8 16 15 31 0; do
  PVCNN_PW_ABLATE=$ab timeout 200 python tools/phase_probe.py pw --shape 16x1472x512x4096 --ablate 2>/dev/null | tee -a $O/ablate_pw_fwd.jsonl
done
