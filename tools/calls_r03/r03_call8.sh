#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_train_parity.py > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json
 PVCNN_GATHER_PIPE=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_20_5_nopipe.json
 timeout 200 python tools/opbench.py 2>/dev/null | grep median > $O/opbench.jsonl)
ls $O
