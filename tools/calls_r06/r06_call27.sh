#!/bin/bash
# round 6, call 27: the round's evidence set again, on the FINAL kernel sources (backward-weight kernels, ABI v12): the whole GPU suite,
# smoke, bench lines, traces, counters, other configs; + the ablation tables of the wide kernels and the backward-weight table
ROUND=r06 bash tools/evidence_call.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; cd $R
make -C tools/probe > $O/probe_make.log 2>&1
: > $O/ablate_pw_fwd.jsonl; : > $O/ablate_conv_fwd.jsonl
for ab in 0 1 2 4 8 16 31; do
  PVCNN_PW_ABLATE=$ab timeout 200 python tools/phase_probe.py pw --shape 16x1472x512x4096 --ablate 2>/dev/null >> $O/ablate_pw_fwd.jsonl
  PVCNN_CONV_ABLATE=$ab timeout 200 python tools/phase_probe.py conv 2>/dev/null >> $O/ablate_conv_fwd.jsonl
done
for pp in 0 1; do PVCNN_WGRAD_PP=$pp timeout 300 python tools/wgrad_only.py 2>/dev/null | sed "s/^{/{\"PVCNN_WGRAD_PP\": $pp, /" >> $O/wgrad_table.jsonl; done
cat $O/ablate_pw_fwd.jsonl $O/ablate_conv_fwd.jsonl $O/wgrad_table.jsonl
