#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03v; mkdir -p $O
cd $R
for e in 0 8 7 15; do
  echo "EXP=$e"; PVCNN_CONV_EXP=$e python tools/convcheck.py --time --no-check --shapes 16x64x64x16,16x128x128x16 2>/dev/null | grep "time_split.*nsplit\": 2" | cut -c1-120
done
