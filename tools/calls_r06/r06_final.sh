#!/bin/bash
# round 6, final state: the whole GPU suite in ONE process as the driver runs it, smoke, then the driver's bench command -- after
# profiles/kernel_durations*.json and pmc_traffic.json of these kernel sources were committed, so the line cites them
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_suite_one_process.log 2>&1; grep -E "passed|failed" $O/gpu_suite_one_process.log | tail -1
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | tee $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_20_5.err | tail -1 > $O/bench_20_5.json
timeout 600 python bench.py --reference-composition --no-cpu-baseline --no-variants --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench_reference_composition.json
python -c "
import json
d=json.load(open('$O/bench_20_5.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'refcomp', d.get('reference_composition_value'), 'fp32', d.get('fp32_mfma_value'), 'roofline', r['frac'], r['priced_on'], 'traffic', r.get('traffic'), r['in_graph_trace'].get('matches_running_sources'), 'mfma', d['roofline_mfma']['frac'], 'cpu', d['cpu_baseline']['value'])
d=json.load(open('$O/bench_reference_composition.json')); print('refcomp line', d['value'], d['ms_per_step'], d['eager_value'], d['config']['composition'])"
