#!/bin/bash
# round 5, call 5: three experiment switches, same box: streaming stores in the Conv3d epilogue (does the gather behind it speed up?),
# a two-deep entry prefetch in the scatter applies, the dense head off; per-kernel in-graph times for the first two
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
PVCNN_SEGSUM_PREFETCH=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plans.py tests/test_gpu_properties.py -q -m gpu -p no:cacheprovider > $O/prefetch_tests.log 2>&1
echo "prefetch tests: $(grep -E 'passed|failed' $O/prefetch_tests.log | tail -1)"
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench_$name.json
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], 'gather live', d['roofline']['avg_us'])"; }
run base A=0
run nt_store PVCNN_CONV_NT_STORE=1
run prefetch PVCNN_SEGSUM_PREFETCH=1
run dense_off PVCNN_DENSE_HEAD=0
run base2 A=0
cd /tmp && export TMPDIR=/tmp
for v in base nt_store prefetch; do
  case $v in base) E="A=0";; nt_store) E="PVCNN_CONV_NT_STORE=1";; prefetch) E="PVCNN_SEGSUM_PREFETCH=1";; esac
  rm -rf /tmp/kt_$v; env $E timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -- python $R/bench.py --no-cpu-baseline --no-variants --steps 40 --warmup 10 > /dev/null 2>&1
  t=$(find /tmp/kt_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_steady.py $t 40 60 20 --by-grid gather_lds > $O/trace_$v.txt 2>&1
  python $R/tools/trace_steady.py $t 40 0 20 --by-grid segsum > $O/trace_segsum_$v.txt 2>&1
  python $R/tools/trace_steady.py $t 40 0 20 --by-grid conv3d_igemm > $O/trace_conv_$v.txt 2>&1
  echo "== $v"; head -2 $O/trace_$v.txt; grep "gather_lds_pipe_kernel" $O/trace_$v.txt | head -2; grep "grid=" $O/trace_segsum_$v.txt | head -8 | cut -c1-110; grep "grid=" $O/trace_conv_$v.txt | head -4 | cut -c1-110
done
