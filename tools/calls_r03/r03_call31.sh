#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03zc; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_gather_pipe.py tests/test_gpu_parity.py tests/test_gpu_bnact.py tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
python tools/opbench.py --ops devox_fwd --shapes 16x64x4096x32,8x32x8192x32,16x9x4096x32,16x128x4096x32 2>/dev/null | grep median | cut -c1-260
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03zc/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], 'roofline', r['frac'], r['avg_us'], r['with_fused_addend']['frac'])
for k in d['kernels']:
    if 'devoxelize_fwd' in k['kernel']: print(k['shape_BCNR'], k['avg_us'], k['frac_of_8TBs'])
PY
