#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -B30 "short test summary" | cut -c1-220 | tail -45
