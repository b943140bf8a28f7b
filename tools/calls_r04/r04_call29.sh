#!/bin/bash
# cfg5 lines of the evidence set again (the fused Frustum box loss changed that step only), and the frustum / as-benched tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
for c in cfg5; do
  rm -rf /tmp/kt_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- $BENCH --config $c --no-cpu-baseline --steps 40 --warmup 10 > $O/${c}_under_rocprof.log 2>&1
  t=$(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1); python $R/tools/trace_steady.py $t 40 50 50 --json $O/kernel_durations_$c.json > $O/${c}_steady_state.txt 2>&1
  cp $O/kernel_durations_$c.json $R/profiles/kernel_durations_$c.json
  (cd $R && timeout 300 python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_$c.json)
done
rm -f $O/*_under_rocprof.log
head -3 $O/cfg5_steady_state.txt
python -c "
import json; d=json.load(open('$O/bench_cfg5.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_mfma',{}).get('frac'))"
(cd $R && timeout 600 python -m pytest tests/test_gpu_frustum_loss.py tests/test_gpu_train_parity.py -q -p no:cacheprovider -k "frustum" 2>&1 | tail -2)
