#!/bin/bash
# round 6, call 22: the zero-row test as a per-strip scalar bit mask (no LDS reads at the top of a step): bit-identity, time, phases
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06v; mkdir -p $O
cd $R
make -C tools/probe libprobe_conv3d_wgrad_f16.so libablate_conv3d_wgrad_f16.so > $O/make.log 2>&1
timeout 900 python -m pytest tests/test_gpu_wgrad_pp.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for pp in 0 1; do
  PVCNN_WGRAD_PP=$pp timeout 300 python tools/wgrad_only.py > $O/wgrad_pp$pp.jsonl 2> $O/wgrad_pp$pp.err; echo "pp=$pp"; cat $O/wgrad_pp$pp.jsonl
done
for ab in 64 16; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --ablate --shapes 16x64x64x16,16x128x128x16,16x64x64x32 >> $O/wgrad_variants.jsonl 2>> $O/wgrad_variants.err
done
cat $O/wgrad_variants.jsonl
for ab in 0 64; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --probe --shapes 16x128x128x16,16x64x64x32 >> $O/wgrad_phases.jsonl 2>> $O/wgrad_phases.err
done
cat $O/wgrad_phases.jsonl
