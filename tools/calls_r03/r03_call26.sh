#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03za; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R; timeout 300 python -m pytest tests/test_gpu_optim.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/log 2>&1
grep "^{\"metric\"" $O/log | tail -1 > $O/bench_under_rocprof.json; rm -f $O/log
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steady.py $t 100 60 43 --by-grid bnact > $O/bench_steady_state.txt 2>&1
python $R/tools/trace_steady.py $t 100 0 43 --by-grid "pw_gemm_kernel" | grep -A20 "by launch grid" > $O/pw_fp32_by_grid.txt 2>&1
grep -A40 "by launch grid" $O/bench_steady_state.txt | cut -c1-170; cat $O/pw_fp32_by_grid.txt | cut -c1-170
