#!/bin/bash
# Conv3d backward-weight, two x planes per step at R <= 16 (XP = 2): parity, timing, bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_range.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_train_parity.py -m gpu -q -x -p no:cacheprovider -k "pvconv" 2>&1 | tail -2 | cut -c1-200
python tools/convcheck.py --time --no-check --shapes 16x64x64x16,16x128x128x16,16x64x128x16,32x64x64x12,8x128x128x8,8x256x256x8,32x64x64x16 2>/dev/null | grep "time_wgrad" | cut -c1-200 | tee $O/wgrad.jsonl
python tools/sparse_probe.py 2>/dev/null | grep "B=" | tee $O/sparse_probe.txt
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$i.json; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['ms_per_step'])"; done
for c in cfg3 cfg5; do timeout 300 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('$c', d['value'], d['ms_per_step'])"; done
