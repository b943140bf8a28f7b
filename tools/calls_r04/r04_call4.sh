#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R && timeout 300 python tools/sparse_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/sparse_probe.txt
