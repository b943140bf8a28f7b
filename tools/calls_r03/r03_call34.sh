#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03ze; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_conv3d.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32 2>/dev/null | grep "wgrad_f16" | cut -c1-160
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r03ze/bench.json')); print(d['value'], d['ms_per_step'])"
