#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R && python tools/eager_profile.py 2>&1 | grep -v amdgpu.ids | head -50 | cut -c1-160 > $O/eager_plain.txt
python tools/eager_profile.py --after-graph 2>&1 | grep -v amdgpu.ids | head -50 | cut -c1-160 > $O/eager_after_graph.txt
head -3 $O/eager_plain.txt; head -45 $O/eager_after_graph.txt
