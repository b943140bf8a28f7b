#!/bin/bash
# Evidence set of a round, one gpurun call (~4 GPU-minutes):  [ROUND=r03] tools/collect_evidence.sh
# Outputs -> gpurun_out/$ROUND/; copy what is judged to profiles/ as ${ROUND}_*  (tools/publish_evidence.sh does that).
R=${GRAFT_REPO_ROOT:-$PWD}; RD=${ROUND:-r03}; O=$R/gpurun_out/$RD; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
# 1. the bench line under the DRIVER's command, and with the default flags (100 steps, 30 warm-ups); cpu_baseline at the full batch
(cd $R && timeout 600 $BENCH --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
(cd $R && timeout 600 $BENCH --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json)
(cd $R && timeout 300 $BENCH --no-cpu-baseline --eager --steps 30 --warmup 10 2>/dev/null | tail -1 > $O/bench_eager.json)
# 2. rocprofv3 kernel trace + stats of the same command -> kernel stats, steady-state reduction of the 100 replayed steps
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $BENCH --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -90 $f > $O/bench_kernel_stats.csv
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 100 90 43 > $O/bench_steady_state.txt 2>&1
# 3. HBM traffic counters, separate passes, ON THE BENCH COMMAND ITSELF (every pvcnn:: kernel as it runs in the replayed step)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- $BENCH --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn --json > $O/pmc_${c}_bench.json
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn > $O/pmc_${c}_bench.txt
done
python $R/tools/make_pmc_traffic.py $O > $O/pmc_traffic.json
# 4. op-level and convolution tables, the other BASELINE configs (+ their steady states)
(cd $R && python tools/opbench.py 2>/dev/null | grep median > $O/opbench.jsonl; python tools/opbench.py --kind surface 2>/dev/null | grep median >> $O/opbench.jsonl
 python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32,16x64x64x16,16x64x128x16,16x128x128x16,32x64x64x12,32x64x64x16,32x64x128x12 2>/dev/null | grep "time_\|absmax" > $O/convbench.jsonl
 python tools/pwbench.py 2>/dev/null | grep "^{" > $O/pwbench.jsonl
 python tools/fpsbench.py 2>/dev/null | grep "^{" > $O/fpsbench.jsonl
 PVCNN_CONV_MATH=fp32 PVCNN_PW_MATH=fp32 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_fp32_mfma.json
 timeout 300 python bench.py --no-cpu-baseline --torch-adam --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_torch_adam.json
 timeout 120 python tools/step_profile.py --rows 70 > $O/step_profile.txt 2>/dev/null)
for c in cfg3 cfg4 cfg5; do
  rm -rf /tmp/kt_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- $BENCH --config $c --no-cpu-baseline --steps 40 --warmup 10 > $O/${c}_under_rocprof.log 2>&1
  t=$(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 43 > $O/${c}_steady_state.txt 2>&1
  (cd $R && timeout 300 python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_$c.json)
done
rm -f $O/*_under_rocprof.log
ls -la $O
