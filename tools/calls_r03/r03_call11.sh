#!/bin/bash
# full GPU suite (incl. the whole train-parity file), then the round's evidence set
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
 timeout 1200 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1; grep "passed\|failed" $O/train_parity.log | tail -2)
ROUND=r03 bash $R/tools/collect_evidence.sh > /dev/null 2>&1
ls $O | head -50
