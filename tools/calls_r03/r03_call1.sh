#!/bin/bash
# Round 3, first GPU call: (1) the graph-replayed bench under the driver's command and the default one, (2) rocprofv3 kernel stats +
# steady state of the SHIPPED path (incl. gather_lds_pipe_kernel), (3) FETCH_SIZE / WRITE_SIZE passes on the same command,
# (4) every opt-in variant measured (keep or delete), (5) torch-glue attribution, (6) the other configs.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
(cd $R && timeout 300 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dp_rccl.py -q -m gpu -p no:cacheprovider > $O/graph_tests.log 2>&1)
(cd $R && timeout 400 $BENCH --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
(cd $R && timeout 400 $BENCH --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json)
(cd $R && timeout 300 $BENCH --no-cpu-baseline --eager --steps 30 --warmup 10 2>/dev/null | tail -1 > $O/bench_eager.json)
# rocprofv3 kernel trace + stats of the same command
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $BENCH --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -90 $f > $O/bench_kernel_stats.csv
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 80 43 > $O/bench_steady_state.txt 2>&1
# PMC passes (separate), on the bench command itself
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- $BENCH --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn --json > $O/pmc_${c}_bench.json
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn > $O/pmc_${c}_bench.txt
done
# variants: keep or delete
(cd $R
 PVCNN_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_experimental.py -q -m gpu -p no:cacheprovider > $O/experimental_tests.log 2>&1
 timeout 200 python tools/pwbench.py --shapes 16x128x1024x4096,16x1472x512x4096,16x512x256x4096 2>/dev/null | grep "^{" > $O/pwbench.jsonl
 PVCNN_PW_MB8=1 timeout 200 python tools/pwbench.py --shapes 16x128x1024x4096,16x1472x512x4096,16x512x256x4096 2>/dev/null | grep "^{" > $O/pwbench_mb8.jsonl
 timeout 200 python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32,16x64x64x16,16x64x128x16,16x128x128x16 2>/dev/null | grep "time_\|absmax" > $O/convbench.jsonl
 PVCNN_WGRAD_REDUCE=2 timeout 120 python tools/convcheck.py --time --no-check --shapes 16x64x64x32,16x64x64x16,16x128x128x16 2>/dev/null | grep "time_wgrad_f16" > $O/convbench_wgrad_reduce2.jsonl
 for v in "PVCNN_PW_MB8=1" "PVCNN_WGRAD_REDUCE=2" "PVCNN_BWD_AMAX=1 PVCNN_AMAX_REDUCE=2" "PVCNN_BWD_AMAX=1" "PVCNN_FOLD_BN=1"; do
   n=$(echo $v | tr ' =' '__'); env $v timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_$n.json
 done
 timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_default_40.json
 timeout 120 python tools/step_profile.py > $O/step_profile.txt 2>/dev/null
 timeout 200 python tools/opbench.py 2>/dev/null | grep median > $O/opbench.jsonl
 for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 10 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done)
ls -la $O
