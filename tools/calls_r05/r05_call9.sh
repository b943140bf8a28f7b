#!/bin/bash
# PVCNN++ (cfg3): furthest-point sampling of the whole pyramid ahead on a stream of its own (a parallel path of the captured graph),
# with and without raised wave priority in the sampling kernel -- tests, A/B on one box, and a kernel trace of what overlaps what.
# Also: the driver's bench command on the GPU-paced live roofline timing (bench.py KernelClock.burst).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_centers_ahead.py tests/test_gpu_graph.py tests/test_gpu_parity_as_benched.py::test_full_width_cfg3_step_pvcnnpp tests/test_gpu_parity.py tests/test_gpu_train_parity.py -k "centers or ahead or hand_off or graph or cfg3 or furthest or fps or PVCNN2 or pvcnnpp" -x -q -p no:cacheprovider > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -2
: > $O/ab.txt
for i in 1 2; do for a in 0 1; do for p in 0 1; do
  PVCNN_CENTERS_AHEAD=$a PVCNN_FPS_PRIO=$p timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-variants --steps 60 --warmup 10 2>$O/err_${a}_$p.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ahead', $a, 'prio', $p, 'cfg3', d['value'], d['ms_per_step'], d.get('eager_value'))" | tee -a $O/ab.txt
done; done; done
timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20_5.json
python -c "
import json; d=json.load(open('$O/bench_20_5.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'roofline', r['frac'], 'burst', r['avg_us'], 'in-step', r['in_step_event_pair_us'], 'in-graph', r['in_graph_us'], r['priced_on'])" | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for a in 1 0; do
  rm -rf /tmp/tr_$a; PVCNN_CENTERS_AHEAD=$a timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$a -- python $R/bench.py --config cfg3 --no-cpu-baseline --no-variants --steps 40 --warmup 5 > /dev/null 2>&1
  f=$(find /tmp/tr_$a -name '*kernel_trace.csv' | head -1)
  echo "== PVCNN_CENTERS_AHEAD=$a" | tee -a $O/overlap.txt
  python $R/tools/trace_overlap.py $f 'fps_kernel<512' 40 | tee -a $O/overlap.txt
  python $R/tools/trace_overlap.py $f 'fps_kernel<64' 40 | tee -a $O/overlap.txt
  python $R/tools/trace_steady.py $f 40 8 50 | head -10 | tee -a $O/overlap.txt
done
