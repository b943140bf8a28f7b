#!/bin/bash
# round 6, call 18: why the ping-pong backward-weight kernel is slower: variants (32 no per-unit scheduling fence, 64 every wave multiplies first) and ablations
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06r; mkdir -p $O
cd $R
make -C tools/probe libablate_conv3d_wgrad_f16.so > $O/make.log 2>&1
for ab in 0 32 64 96 1 2 4 16 36 100; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --ablate --shapes 16x128x128x16,16x64x64x32 >> $O/wgrad_variants.jsonl 2>> $O/wgrad_variants.err
done
cat $O/wgrad_variants.jsonl
