#!/bin/bash
# Multi-GPU readiness check for a node with >= 2 MI355X (the build box has one GPU, so this has never run here).
#   tools/scale_check.sh [N ...]          default: 1 2 4 8 (those <= the GPUs present)
# For every N: the driver's bench command, then asserts on the JSON line
#   rccl_ranks == N (N > 1), the step is ONE graph launch with the collectives inside ('graph+collectives'; the fallback is reported
#   with its reason and fails the check), per-rank ms/step spread < 5 %, and prints allreduce_alone_us_per_step next to the ring
#   estimate 2 (N-1)/N x gradient_bytes / 153 GB/s per xGMI link (0.19 ms at 16 MiB on 8 GPUs).
# Exit code 0 only if every N passed.  Outputs: gpurun_out/scale_check/N<k>.json + summary.txt
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/scale_check; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
Ns=${@:-1 2 4 8}; rc=0
# the sequencing without hardware first (gloo / CPU): both orderings, lock-step, == big-batch SGD
python $R/bench.py --gpus 4 --dry-collectives 2>/dev/null | tail -1 > $O/dry_collectives.json || rc=1
: > $O/summary.txt
for N in $Ns; do
  if [ "$N" -gt "$have" ]; then echo "N=$N skipped: $have GPU(s) present" | tee -a $O/summary.txt; continue; fi
  (cd $R && timeout 900 python bench.py --gpus $N --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>$O/N$N.err | tail -1 > $O/N$N.json)
  python - $N $O/N$N.json <<'PY' | tee -a $O/summary.txt || rc=1
import json, sys
n, path = int(sys.argv[1]), sys.argv[2]
try:
    d = json.load(open(path))
except Exception as exc:
    print(f'N={n} FAILED: no JSON line ({exc})'); sys.exit(1)
c = d['config']; ok = True; notes = []
if n > 1 and c['rccl_ranks'] != n:
    ok = False; notes.append(f"rccl_ranks {c['rccl_ranks']} != {n}")
if n > 1 and c.get('step_mode') != 'graph+collectives':
    ok = False; notes.append('step mode: ' + str(c.get('step_mode')) + ' / ' + str(d.get('collective_capture_error') or d.get('graph_error')))
ms = c['per_rank_ms_per_step']; spread = (max(ms) - min(ms)) / min(ms)
if spread >= 0.05:
    ok = False; notes.append(f'per-rank spread {spread:.1%}')
ring_us = 2 * (n - 1) / n * c['gradient_bytes'] / 153e9 * 1e6 if n > 1 else 0.0
print(f"N={n} {'ok' if ok else 'FAILED'}: {d['value']:.1f} {d['unit']} ({d['ms_per_step']} ms/step, per rank {ms}); buckets {c['gradient_buckets']}, "
      f"all-reduce alone {c['allreduce_alone_us_per_step']} us/step vs ring estimate {ring_us:.0f} us; " + '; '.join(notes))
sys.exit(0 if ok else 1)
PY
done
python - $O <<'PY' | tee -a $O/summary.txt
import glob, json, os, sys
vals = {}
for p in glob.glob(os.path.join(sys.argv[1], 'N*.json')):
    try:
        d = json.load(open(p)); vals[d['n_gpus']] = d['value']
    except Exception:
        pass
if 1 in vals:
    print('weak-scaling efficiency vs N=1: ' + ', '.join(f'N={n}: {vals[n] / (n * vals[1]):.3f}' for n in sorted(vals)))
PY
exit $rc
