#!/bin/bash
# round 6, call 23: the backward-weight kernel with the live step interleaved by hand (one pinned piece of conversion / request behind every MFMA):
# test, per-layer time next to the old kernel and to the un-interleaved form (ablate 128), the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06w; mkdir -p $O
cd $R
make -C tools/probe libprobe_conv3d_wgrad_f16.so libablate_conv3d_wgrad_f16.so > $O/make.log 2>&1
timeout 900 python -m pytest tests/test_gpu_wgrad_pp.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for pp in 0 1; do
  PVCNN_WGRAD_PP=$pp timeout 300 python tools/wgrad_only.py > $O/wgrad_pp$pp.jsonl 2> $O/wgrad_pp$pp.err; echo "pp=$pp"; cat $O/wgrad_pp$pp.jsonl
done
for ab in 128 16; do
  PVCNN_WGRAD_ABLATE=$ab timeout 300 python tools/wgrad_only.py --ablate --shapes 16x64x64x16,16x128x128x16,16x64x64x32 >> $O/wgrad_variants.jsonl 2>> $O/wgrad_variants.err
done
cat $O/wgrad_variants.jsonl
for pp in 1 0 1 0; do
  PVCNN_WGRAD_PP=$pp timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err
  python - <<PY
import json; d=json.loads(open('$O/bench_pp$pp.json').read().strip().splitlines()[-1]); print('pp=$pp', d['value'], d['ms_per_step'])
PY
done
