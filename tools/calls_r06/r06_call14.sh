#!/bin/bash
# round 6, call 14: in-graph durations of the 1x1 GEMM launches with 512 x 128 items (PVCNN_PW_WIDE=1) and with 256 x 256 items (=2)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 1 2; do
  rm -rf /tmp/kt$w; PVCNN_PW_WIDE=$w timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$w -- python $R/bench.py --no-cpu-baseline --no-variants --steps 40 --warmup 10 > $O/bench_rocprof_wide$w.log 2>&1
  t=$(find /tmp/kt$w -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_steady.py $t 40 90 50 --by-grid pw_gemm > $O/steady_wide$w.txt 2>&1
  grep -A12 'kernels matching "pw_gemm"' $O/steady_wide$w.txt | cut -c1-150
  head -3 $O/steady_wide$w.txt
done
