#!/bin/bash
# Evidence set of a round, one gpurun call (~6 GPU-minutes):  [ROUND=r05] [QUICK=1] tools/collect_evidence.sh
# Outputs -> gpurun_out/$ROUND/; copy what is judged to profiles/ as ${ROUND}_*  (tools/publish_evidence.sh does that).
R=${GRAFT_REPO_ROOT:-$PWD}; RD=${ROUND:-r05}; O=$R/gpurun_out/$RD; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
# 1. rocprofv3 kernel trace + stats of the bench command -> kernel stats, steady-state reduction of the 100 replayed steps, and the
#    per-launch in-graph durations bench.py prices its rooflines on (profiles/kernel_durations.json): FIRST, so that the bench lines
#    of step 2 already carry this round's in-graph figures
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $BENCH --no-cpu-baseline --no-variants > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -90 $f > $O/bench_kernel_stats.csv
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steady.py $t 100 90 50 --by-grid conv3d_igemm --json $O/kernel_durations.json > $O/bench_steady_state.txt 2>&1
python $R/tools/trace_steady.py $t 100 0 50 --by-grid gather_lds > $O/bench_gather_by_launch.txt 2>&1
python $R/tools/trace_steady.py $t 100 0 50 --by-grid bnact > $O/bench_bnact_by_launch.txt 2>&1
cp $O/kernel_durations.json $R/profiles/kernel_durations.json
# 2. the bench line under the DRIVER's command, and with the default flags (100 steps, 30 warm-ups); cpu_baseline at the full batch
(cd $R && timeout 600 $BENCH --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json)
(cd $R && timeout 600 $BENCH --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json)
(cd $R && timeout 300 $BENCH --no-cpu-baseline --no-variants --eager --steps 30 --warmup 10 2>/dev/null | tail -1 > $O/bench_eager.json)
(cd $R && timeout 300 $BENCH --no-cpu-baseline --no-variants --eager --torch-adam --steps 30 --warmup 10 2>/dev/null | tail -1 > $O/bench_eager_torch_adam.json)
if [ -z "$QUICK" ]; then     # QUICK=1: the bench lines, the traces and the other configs only (the counters of steps 3-5 belong to kernels that did not change)
# 3. HBM traffic counters, separate passes, ON THE BENCH COMMAND ITSELF, per (kernel, launch grid)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- $BENCH --steps 10 --warmup 3 --no-cpu-baseline --no-variants > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn --json --by-grid > $O/pmc_${c}_bench.json
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn --by-grid > $O/pmc_${c}_bench.txt
done
#    ... and on known byte counts (the calibration of the counters in these kernels' access patterns)
: > $O/pmc_calibration.jsonl
for spec in devox_fwd:16x64x4096x16 devox_fwd:16x64x4096x32 devox_bwd_apply:16x64x4096x32 vox_apply:16x64x4096x32 vox_bwd:16x64x4096x16; do
  op=${spec%%:*}; sh=${spec##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal_$c; timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -- python $R/tools/opbench.py --ops $op --shapes $sh --iters 3 > /dev/null 2>&1
    python $R/tools/pmc_by_kernel.py /tmp/cal_$c pvcnn --json --by-grid > /tmp/cal_$c.json
  done
  python $R/tools/pmc_calibration.py $op $sh /tmp/cal_FETCH_SIZE.json /tmp/cal_WRITE_SIZE.json >> $O/pmc_calibration.jsonl
done
python $R/tools/make_pmc_traffic.py $O $O/bench_20_5.json > $O/pmc_traffic.json
# 4. MFMA utilisation counters of the matrix kernels inside the bench step, with the effective clock; the same on the fill probe
S1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; S2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for i in 1 2; do
  eval C=\$S$i
  rm -rf /tmp/mf$i; timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/mf$i -- $BENCH --steps 10 --warmup 3 --no-cpu-baseline --no-variants > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/mf$i pvcnn --json --by-grid > $O/pmc_mfma_set${i}_bench.json
done
python $R/tools/pmc_mfma_table.py $O/pmc_mfma_set1_bench.json $O/pmc_mfma_set2_bench.json > $O/pmc_mfma_bench_table.md
#    the power claim as data: the SAME binaries on random and on constant operands, one fill per process -> duration, effective clock, pipe busy
: > $O/pmc_mfma_fill_probe.jsonl
for kind in randn const; do
  rm -rf /tmp/fp_$kind; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/fp_$kind -- python $R/tools/mfma_fill_probe.py --kinds $kind > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/fp_$kind pvcnn --json --by-grid | python -c "
import json, sys
for k, v in sorted(json.load(sys.stdin).items()):
    if v.get('SQ_INSTS_MFMA', 0) > 5e5 and v.get('GRBM_GUI_ACTIVE') and v.get('avg_us'):
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        print(json.dumps({'fill': '$kind', 'kernel': k.replace('void ', '').split('(')[0] + ' @' + k.split('@grid=')[-1], 'launches': v['dispatches'],
                          'us_under_pmc': round(v['avg_us'], 1), 'effective_clock_GHz': round(cyc / v['avg_us'] / 1e3, 3),
                          'mfma_pipe_busy': round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), 3), 'shader_cycles': round(cyc)}))" >> $O/pmc_mfma_fill_probe.jsonl
done
(cd $R && timeout 200 python tools/mfma_fill_probe.py > $O/mfma_fill_probe.txt 2>&1)
# 5. op-level and convolution tables, the other BASELINE configs (+ their steady states)
(cd $R && python tools/opbench.py 2>/dev/null | grep median > $O/opbench.jsonl; python tools/opbench.py --kind surface 2>/dev/null | grep median >> $O/opbench.jsonl
 python tools/opbench.py --ops devox_fwd --shapes 8x64x2048x32,16x128x4096x32 2>/dev/null | grep median >> $O/opbench.jsonl
 python tools/convcheck.py --time --no-check --shapes 16x9x64x32,16x64x64x32,16x64x64x16,16x64x128x16,16x128x128x16,32x64x64x12,32x64x64x16,32x64x128x12 2>/dev/null | grep "time_\|absmax" > $O/convbench.jsonl
 python tools/pwbench.py 2>/dev/null | grep "^{" > $O/pwbench.jsonl
 python tools/fpsbench.py 2>/dev/null | grep "^{" > $O/fpsbench.jsonl
 python tools/sparse_probe.py 2>/dev/null | grep "B=" > $O/sparse_probe.txt
 PVCNN_CONV_MATH=fp32 PVCNN_PW_MATH=fp32 timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_fp32_mfma.json
 timeout 120 python tools/step_profile.py --rows 70 > $O/step_profile.txt 2>/dev/null)
fi
for c in cfg3 cfg4 cfg5; do
  rm -rf /tmp/kt_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- $BENCH --config $c --no-cpu-baseline --no-variants --steps 40 --warmup 10 > $O/${c}_under_rocprof.log 2>&1
  t=$(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 50 --json $O/kernel_durations_$c.json > $O/${c}_steady_state.txt 2>&1
  cp $O/kernel_durations_$c.json $R/profiles/kernel_durations_$c.json
  (cd $R && timeout 300 python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_$c.json)
done
rm -f $O/*_under_rocprof.log
ls -la $O
