#!/bin/bash
# Evidence set, one gpurun call (~6 GPU-minutes):  [ROUND=r03] tools/collect_evidence.sh   (outputs -> gpurun_out/$ROUND/, copy what is judged to
# profiles/ as ${ROUND}_*; tools/collect_evidence_short.sh is the 90-second subset: bench + rocprof stats + steady state)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${ROUND:-r03}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
# 1. the bench line (default flags: 1 GPU, 100 steps, 30 warm-up; includes cpu_baseline at the full batch)
(cd $R && timeout 600 $BENCH 2>$O/bench.err | tail -1 > $O/bench.json)
# 2. rocprofv3 kernel trace + stats of the same command (no CPU baseline) -> kernel stats, steady-state reduction
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $BENCH --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -80 $f > $O/bench_kernel_stats.csv
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 70 20 > $O/bench_steady_state.txt
# 3. HBM traffic counters, separate passes, ON THE BENCH COMMAND ITSELF (the fused gather as it runs in the step) ...
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- $BENCH --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn --json > $O/pmc_${c}_bench.json
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn > $O/pmc_${c}_bench.txt
done
# ... and per op at the devoxelize-backward shapes (one shape per run so that launches of one kernel template are one shape)
for shp in 16x64x4096x16 16x128x4096x16 16x64x4096x32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmco; timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmco -- python $R/tools/opbench.py --ops devox_bwd_apply,devox_fwd --shapes $shp --iters 3 > /dev/null 2>&1
    python $R/tools/pmc_by_kernel.py /tmp/pmco pvcnn --json > $O/pmc_${c}_opbench_$shp.json
  done
done
python $R/tools/make_pmc_traffic.py $O > $O/pmc_traffic.json
# 4. op-level and convolution tables, the other BASELINE configs
(cd $R && python tools/opbench.py 2>/dev/null | grep median > $O/opbench.jsonl; python tools/opbench.py --kind surface 2>/dev/null | grep median >> $O/opbench.jsonl
 python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32,16x64x64x16,16x64x128x16,16x128x64x16,16x128x128x16,32x64x64x12,32x64x64x16,32x64x128x12 2>/dev/null | grep "time_\|absmax" > $O/convbench.jsonl
 python tools/pwbench.py 2>/dev/null | grep "^{" > $O/pwbench.jsonl
 PVCNN_CONV_MATH=fp32 PVCNN_PW_MATH=fp32 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fp32_mfma.json
 PVCNN_CONV_MATH=bf16x3 PVCNN_PW_MATH=fp32 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16x3.json
 for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$c.json; done)
# 5. opt-in / environment-selected kernel variants: bit-identity across processes, kernel-level A/B
(cd $R && PVCNN_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_experimental.py -q -m gpu -p no:cacheprovider > $O/experimental_tests.log 2>&1
 timeout 120 python tools/foldbench.py 2>/dev/null | grep "^{" > $O/foldbench.jsonl
 PVCNN_AMAX_REDUCE=2 timeout 120 python tools/foldbench.py 2>/dev/null | grep "^{" > $O/foldbench_amax_form2.jsonl
 PVCNN_GATHER_PIPE=0 timeout 60 python tools/pipecheck.py 2>/dev/null | grep "^{" > $O/pipecheck.jsonl
 timeout 60 python tools/pipecheck.py 2>/dev/null | grep "^{" >> $O/pipecheck.jsonl
 PVCNN_WGRAD_REDUCE=2 timeout 120 python tools/convcheck.py --time --no-check --shapes 16x64x64x32,16x64x64x16,16x128x128x16 2>/dev/null | grep "time_wgrad_f16" > $O/convbench_wgrad_reduce2.jsonl
 PVCNN_PW_MB8=1 timeout 120 python tools/pwbench.py 2>/dev/null | grep "^{" > $O/pwbench_mb8.jsonl
 PVCNN_PW_MB8=1 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pw_mb8.json
 PVCNN_BWD_AMAX=1 PVCNN_AMAX_REDUCE=2 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_amax_form2.json
 timeout 120 python tools/step_profile.py > $O/step_profile.txt 2>/dev/null
 timeout 200 python bench.py --graph --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_with_graph_replay.json)
ls -la $O
