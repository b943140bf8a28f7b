#!/bin/bash
# threshold of the split 1x1 path: 2^32 (round 2) vs 2^30 / 2^29 / 2^28 macs; and which layers run pw_gemm_kernel<2,false,false>
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for t in 27 26 24 1; do
  for c in cfg2 cfg3 cfg4 cfg5; do
    extra=""; [ $c != cfg2 ] && extra="--config $c --steps 40 --warmup 10"
    PVCNN_PW_SPLIT_MIN_LOG2=$t timeout 300 python bench.py --no-cpu-baseline $extra 2>/dev/null | tail -1 > $O/bench_${c}_t$t.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03w/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('eager_value'))
    except Exception as e: print(f, 'ERR', e)
PY
