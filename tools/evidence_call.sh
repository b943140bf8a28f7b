#!/bin/bash
# full GPU suite (incl. the whole train-parity files), then the round's evidence set
R=${GRAFT_REPO_ROOT:-$PWD}; RD=${ROUND:-r04}; O=$R/gpurun_out/$RD; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py --deselect tests/test_gpu_parity_as_benched.py > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
 timeout 1200 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_parity_as_benched.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1; grep "passed\|failed" $O/train_parity.log | tail -2)
(cd $R && timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | tee $O/smoke.log)
ROUND=$RD bash $R/tools/collect_evidence.sh > $O/collect.log 2>&1
ls $O | head -70
