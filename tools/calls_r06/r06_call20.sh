#!/bin/bash
# round 6, call 20: is the backward-weight kernel's matrix phase bound by its LDS reads?  variants: 64 all waves multiply first (= the old order),
# +128 without the two halo dwords of a window, +256 one A fragment per k-step, +32 no per-unit fence
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06t; mkdir -p $O
cd $R
make -C tools/probe libablate_conv3d_wgrad_f16.so > $O/make.log 2>&1
for ab in 64 192 448 480; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --ablate --shapes 16x64x64x16,16x128x128x16,16x64x64x32 >> $O/wgrad_variants.jsonl 2>> $O/wgrad_variants.err
done
cat $O/wgrad_variants.jsonl
