#!/bin/bash
# the fixed cost of a fork / join inside a replayed hipGraph (behind profiles/ab/r05j, r05k): a microbenchmark, no product code
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
for a in "200 10" "200 2" "400 20"; do timeout 120 python tools/graph_fork_cost_probe.py $a 2>&1 | tee -a $O/fork_cost.txt; echo | tee -a $O/fork_cost.txt; done
timeout 120 python tools/graph_concurrency_probe.py 2>&1 | tee -a $O/fork_cost.txt
