#!/bin/bash
# wgrad with the rows requested one step ahead: parity, timing, bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py tests/test_gpu_graph.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
python tools/convcheck.py --time --no-check --shapes 16x64x64x32,16x64x64x16,16x128x128x16,16x9x64x32,32x64x64x12,8x128x128x8 2>/dev/null | grep "time_wgrad" | cut -c1-200 | tee $O/wgrad.jsonl
python tools/sparse_probe.py 2>/dev/null | grep "B=" | tee $O/sparse_probe.txt
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_20_5_$i.json; python -c "
import json; d=json.load(open('$O/bench_20_5_$i.json')); print(d['value'], d['ms_per_step'], d['eager_value'])"; done
