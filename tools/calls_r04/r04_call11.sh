#!/bin/bash
# EXPERIMENT: channel rows per segsum_tile workgroup (G = 4 default vs 2): more, smaller workgroups vs fewer re-reads of the plan
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
for g in 0 2; do
  echo "G_EXP=$g"
  PVCNN_SEGSUM_G_EXP=$g python tools/opbench.py --ops vox_apply,devox_bwd_apply --shapes 16x64x4096x16,16x128x4096x16,16x64x4096x32,16x9x4096x32,8x64x2048x32,32x64x1024x12 2>/dev/null | grep median | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['op'], d['BCNR'], d['median_us'], d['frac_8TBs'])" | tee $O/g$g.txt
done
for g in 0 2; do PVCNN_SEGSUM_G_EXP=$g timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d=json.load(sys.stdin); print('bench G_EXP=$g', d['value'], d['ms_per_step'])"; done
