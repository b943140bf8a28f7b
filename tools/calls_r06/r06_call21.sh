#!/bin/bash
# round 6, call 21: coarse phase clocks of the backward-weight step (five stamps per step and wave), ping-pong and multiply-first order
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06u; mkdir -p $O
cd $R
make -C tools/probe libprobe_conv3d_wgrad_f16.so > $O/make.log 2>&1
for ab in 0 64; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --probe --shapes 16x128x128x16,16x64x64x32 >> $O/wgrad_phases.jsonl 2>> $O/wgrad_phases.err
done
cat $O/wgrad_phases.jsonl; tail -3 $O/wgrad_phases.err
