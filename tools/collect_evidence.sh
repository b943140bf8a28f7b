R=$PWD; O=$R/gpurun_out/r1f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. un-profiled bench line (includes cpu_baseline)
(cd $R && timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench.json)
# 2. kernel trace of the same command (no cpu baseline) -> stats + steady-state reduction
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -60 $f > $O/bench_kernel_stats.csv
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 20 60 > $O/bench_steady_state.txt
# 3. PMC passes (separate), op-level shape of the headline kernel
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/tools/opbench.py --ops devox_fwd,vox_fwd,devox_bwd,vox_bwd --shapes 16x64x4096x32 > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$c pvcnn > $O/pmc_${c}_opbench_16x64x4096x32.txt
done
# 4. op-level table
(cd $R && python tools/opbench.py 2>/dev/null | grep median > $O/opbench.jsonl; python tools/opbench.py --kind surface 2>/dev/null | grep median >> $O/opbench.jsonl; python tools/convcheck.py --time --no-check 2>/dev/null | grep time_ > $O/convbench.jsonl)
ls -la $O
