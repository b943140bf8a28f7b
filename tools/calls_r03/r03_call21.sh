#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for e in 0 7; do echo "EXP=$e"; PVCNN_CONV_EXP=$e python tools/mfma_fill_probe.py 2>&1 | tail -9; done
