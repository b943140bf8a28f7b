#!/bin/bash
# round 6, call 15: the round's evidence set on the final kernel sources: the whole GPU suite, smoke, bench lines, traces, counters, other configs
ROUND=r06 bash tools/evidence_call.sh
# + the ablation tables of the two new kernels (tools/probe builds) and the counters of the wide 1x1 GEMM next to the 128-row kernel
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; cd $R
: > $O/ablate_pw_fwd.jsonl; : > $O/ablate_conv_fwd.jsonl
for ab in 0 1 2 4 8 16 31; do
  PVCNN_PW_ABLATE=$ab timeout 200 python tools/phase_probe.py pw --shape 16x1472x512x4096 --ablate 2>/dev/null >> $O/ablate_pw_fwd.jsonl
  PVCNN_CONV_ABLATE=$ab timeout 200 python tools/phase_probe.py conv 2>/dev/null >> $O/ablate_conv_fwd.jsonl
done
cat $O/ablate_pw_fwd.jsonl $O/ablate_conv_fwd.jsonl
