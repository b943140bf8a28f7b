#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 kernel-trace CSV.

usage: trace_steady.py <kernel_trace.csv> <timed_steps> [top] [skip_last] [--by-grid <substring>] [--json <out.json>]
(--by-grid: the kernels whose name contains <substring>, broken down by launch grid AND by position inside the step -- one line per
 launch of a step, i.e. per layer: launches of one template with equal grids (Conv3d 9->64 and 64->64 at 32^3) are told apart by order)
(--json: machine-readable per-launch-of-a-step averages of every pvcnn:: kernel -- profiles/kernel_durations.json, read by bench.py)
Training steps are delimited by the optimizer phase (runs of fused Adam's multi_tensor_apply kernels, or of pvcnn_amd.optim.FlatAdam's kernels);
the `timed_steps` complete steps before the last `skip_last` ones are aggregated, i.e. bench.py's timed region (bench.py runs
min(steps, 20) fully instrumented steps AFTER it: skip_last = 20 for the default 100 steps)."""
import csv
import sys
from collections import defaultdict

by_grid = json_out = None
if '--by-grid' in sys.argv:
    i = sys.argv.index('--by-grid'); by_grid = sys.argv[i + 1]; del sys.argv[i:i + 2]
if '--json' in sys.argv:
    i = sys.argv.index('--json'); json_out = sys.argv[i + 1]; del sys.argv[i:i + 2]
path, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = list(csv.DictReader(open(path)))
name_k = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
s_k = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else 'Start'
e_k = 'End_Timestamp' if 'End_Timestamp' in rows[0] else 'End'
rows.sort(key=lambda r: int(r[s_k]))
# the optimizer's own launches: the gradient packing of pvcnn_amd/dp.py is a multi_tensor_apply kernel too (a copy functor)
is_opt = [('multi_tensor_apply' in r[name_k] and ('FusedOptimizer' in r[name_k] or 'Adam' in r[name_k])) or 'pvcnn::adam_' in r[name_k]
          for r in rows]      # torch's fused Adam, or pvcnn_amd.optim.FlatAdam (adam_flat_kernel ... adam_step_inc_kernel)
ends = [i for i in range(len(rows)) if is_opt[i] and (i + 1 == len(rows) or not is_opt[i + 1])]   # last kernel of each optimizer phase
if len(ends) < steps + 1 + skip:
    raise SystemExit(f'only {len(ends)} optimizer phases in the trace, need {steps + 1 + skip}')
lo, hi = ends[-steps - 1 - skip] + 1, ends[-1 - skip] + 1
steady = rows[lo:hi]
agg = defaultdict(lambda: [0, 0])
for r in steady:
    a = agg[r[name_k]]
    a[0] += 1
    a[1] += int(r[e_k]) - int(r[s_k])
span = (int(steady[-1][e_k]) - int(steady[0][s_k])) / 1e6
busy = sum(v[1] for v in agg.values()) / 1e6
print(f'last {steps} steps: {len(steady)} kernel launches ({len(steady) / steps:.0f}/step), wall {span / steps:.3f} ms/step, '
      f'kernel time {busy / steps:.3f} ms/step ({100 * busy / span:.1f}% GPU busy)')
mine = sum(v[1] for k, v in agg.items() if 'pvcnn::' in k) / 1e6
print(f'hand-written pvcnn:: kernels: {mine / steps:.3f} ms/step ({100 * mine / busy:.1f}% of kernel time)')
for name, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%8.3f ms/step %5.1f%%  calls/step=%6.1f avg=%8.1f us  %s' % (ns / 1e6 / steps, 100 * ns / 1e6 / busy, calls / steps, ns / 1e3 / calls, name[:110]))

# per launch of a step: (kernel name, grid, n-th launch of that (name, grid) inside its step)
gk = [k for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z') if k in rows[0]]
step_ends = set(ends[-steps - 1 - skip:-1 - skip] if skip else ends[-steps - 1:-1])
per_launch = defaultdict(lambda: [0, 0])
seen = defaultdict(int)
for idx in range(lo, hi):
    r = rows[idx]
    key = (r[name_k], tuple(int(r[k]) for k in gk))
    a = per_launch[(key[0], key[1], seen[key])]
    a[0] += 1
    a[1] += int(r[e_k]) - int(r[s_k])
    seen[key] += 1
    if idx in step_ends or is_opt[idx] and (idx + 1 == len(rows) or not is_opt[idx + 1]):
        seen.clear()                                    # the optimizer phase closes a step
if by_grid:
    print(f'--- kernels matching "{by_grid}", one line per launch of a step: grid (threads), position among the equal launches ---')
    for (name, grid, nth), (calls, ns) in sorted(per_launch.items(), key=lambda kv: -kv[1][1]):
        if by_grid in name:
            print('%8.3f ms/step  steps=%4d avg=%8.1f us  grid=%s #%d  %s' % (ns / 1e6 / steps, calls, ns / 1e3 / calls, grid, nth, name[:70]))
if json_out:
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from pvcnn_amd._lib import sources_digest
        digest = sources_digest()
    except Exception:                                   # noqa: BLE001
        digest = None
    out = {'sources_digest': digest, 'trace_commit': os.environ.get('PVCNN_TRACE_COMMIT'),
           'source': 'rocprofv3 --kernel-trace of the bench command, steady-state steps inside the replayed graph (tools/trace_steady.py)',
           'steps': steps, 'wall_ms_per_step': round(span / steps, 4), 'kernel_ms_per_step': round(busy / steps, 4),
           'launches': [{'kernel': name, 'grid': list(grid), 'nth_in_step': nth, 'steps': calls, 'avg_us': round(ns / 1e3 / calls, 3)}
                        for (name, grid, nth), (calls, ns) in sorted(per_launch.items(), key=lambda kv: -kv[1][1]) if 'pvcnn::' in name]}
    json.dump(out, open(json_out, 'w'), indent=0)
