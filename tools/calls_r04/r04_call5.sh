#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R && timeout 300 python tools/sparse_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/sparse_probe.txt
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
python tools/convcheck.py --time --no-check --shapes 16x64x64x32,16x64x64x16,16x128x128x16,16x9x64x32 2>/dev/null | grep "time_" | cut -c1-400 | tee $O/convbench.jsonl
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_20_5_$i.json; python -c "
import json; d=json.load(open('$O/bench_20_5_$i.json')); print(d['value'], d['ms_per_step'])"; done
