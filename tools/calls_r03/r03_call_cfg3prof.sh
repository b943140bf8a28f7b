set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03zf; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fps" 2>&1 | tail -3
cd /tmp
for c in cfg3 cfg4; do
  rm -rf /tmp/kt_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- python $R/bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 > $O/${c}_under_rocprof.log 2>&1
  t=$(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 70 43 > $O/${c}_steady_state.txt 2>&1
done
rm -f $O/*_under_rocprof.log
