#!/bin/bash
# stress the RCCL-in-graph capture: the one-rank test 12 times, each in a fresh process
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
ok=0; bad=0
for i in $(seq 1 12); do
  if timeout 300 python -m pytest tests/test_gpu_dp_rccl.py -m gpu -q -x -p no:cacheprovider > $O/run_$i.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); tail -5 $O/run_$i.log | cut -c1-200; fi
done
echo "rccl capture stress: ok=$ok bad=$bad" | tee $O/summary.txt
