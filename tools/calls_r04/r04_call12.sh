#!/bin/bash
# the whole GPU suite in ONE process, twice (what the driver runs at round end): flakiness check after the capture-mode fix
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04k; mkdir -p $O
cd $R
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/suite_$i.log 2>&1; tail -2 $O/suite_$i.log | cut -c1-200; done
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
