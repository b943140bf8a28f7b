#!/bin/bash
# round 6, call 16: what the Conv3d backward-weight kernel spends on what (ablation builds), R = 16 and R = 32
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06p; mkdir -p $O
cd $R
make -C tools/probe libablate_conv3d_wgrad_f16.so > $O/make.log 2>&1
timeout 300 python tools/wgrad_only.py > $O/wgrad_product.jsonl 2> $O/wgrad_product.err; cat $O/wgrad_product.jsonl; tail -3 $O/wgrad_product.err
for ab in 0 1 2 3 4 8 16 27 31; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --ablate --shapes 16x64x64x16,16x128x128x16,16x64x64x32 >> $O/wgrad_ablate.jsonl 2>> $O/wgrad_ablate.err
done
cat $O/wgrad_ablate.jsonl; tail -3 $O/wgrad_ablate.err
