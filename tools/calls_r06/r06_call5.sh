#!/bin/bash
# round 6, call 5: wide 1x1 GEMM, second version (branch-free request streams, one request per MFMA): bit-identity, speed, phase clocks
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pw_wide.py tests/test_gpu_pwconv.py tests/test_gpu_range.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -5 $O/tests.log
for w in 1 0; do
  PVCNN_PW_WIDE=$w timeout 600 python tools/pwbench.py --shapes 16x1472x512x4096,16x128x1024x4096,16x512x256x4096 2>&1 | grep '"f16x2_BCiCoN"' > $O/pwbench_wide$w.jsonl
  cat $O/pwbench_wide$w.jsonl | cut -c1-200
done
timeout 300 python tools/phase_probe.py pw --shape 16x1472x512x4096 > $O/phase_pw_fwd.json 2> $O/phase_pw_fwd.txt; cat $O/phase_pw_fwd.txt; cut -c1-500 $O/phase_pw_fwd.json
for w in 1 0 1 0; do
  PVCNN_PW_WIDE=$w timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  python -c "import json;d=json.loads(open('$O/bench_wide$w.json').read().strip().splitlines()[-1]);print('wide=$w',d['value'],d['ms_per_step'])"
done
