#!/bin/bash
# round 6, call 13: the wide 1x1 GEMM with 512 x 128 items: bit-identity (three processes), speed, ablation, the step with 512 x 128 / 256 x 256 / 128-row
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pw_wide.py tests/test_gpu_pwconv.py tests/test_gpu_range.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -5 $O/tests.log
for w in 1 2 0; do
  PVCNN_PW_WIDE=$w timeout 600 python tools/pwbench.py --shapes 16x1472x512x4096,16x128x1024x4096,16x512x256x4096 2>&1 | grep '"f16x2_BCiCoN"' > $O/pwbench_wide$w.jsonl
  cat $O/pwbench_wide$w.jsonl | cut -c1-150
done
for ab in 0 2 4 16 31; do
  PVCNN_PW_ABLATE=$ab timeout 200 python tools/phase_probe.py pw --shape 16x1472x512x4096 --ablate 2>/dev/null | tee -a $O/ablate_pw_fwd.jsonl
done
for w in 1 2 0 1 2 0; do
  PVCNN_PW_WIDE=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  python -c "import json;d=json.loads(open('$O/bench_wide$w.json').read().strip().splitlines()[-1]);print('wide=$w',d['value'],d['ms_per_step'])"
done
