#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --single-rank-collectives 2>/dev/null | tail -1 | cut -c1-120
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'])"
