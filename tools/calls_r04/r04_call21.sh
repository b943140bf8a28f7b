#!/bin/bash
# row maxima from the BatchNorm + ReLU pass (ABI v9) and the sample-major BatchNorm traversal: tests, then the bench line and its trace
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bnact.py tests/test_gpu_parity_as_benched.py tests/test_gpu_graph.py tests/test_gpu_train_parity.py -x -q -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steady.py $t 100 70 50 > $O/steady.txt 2>&1
python $R/tools/trace_steady.py $t 100 0 50 --by-grid bnact > $O/bnact.txt 2>&1
head -3 $O/steady.txt
