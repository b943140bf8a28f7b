#!/usr/bin/env python3
"""What ran WHILE a kernel ran: from a rocprofv3 kernel-trace CSV, for every launch of the kernels whose name contains <needle>
(default: the furthest-point sampling of PVCNN++'s first level, `fps_kernel<512`), its duration and the other kernels whose
[start, end] intersects its own -- the evidence for pvcnn_amd.workload.centers_ahead (does the side path of the graph run next to
the convolutions, and what does the chain of M - 1 dependent steps cost there against running alone).

usage: trace_overlap.py <kernel_trace.csv> [needle] [last_n]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else 'fps_kernel<512'
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(path)))
name_k = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
s_k = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else 'Start'
e_k = 'End_Timestamp' if 'End_Timestamp' in rows[0] else 'End'
ev = sorted((int(r[s_k]), int(r[e_k]), r[name_k]) for r in rows)
mine = [e for e in ev if needle in e[2]][-last_n:]
if not mine:
    raise SystemExit(f'no launch of a kernel named *{needle}* in {path}')
starts = [e[0] for e in ev]
import bisect
durs, alone, shared_ns, n_other = [], 0, defaultdict(int), []
for s, e, _ in mine:
    lo = bisect.bisect_left(starts, s - 5_000_000)      # nothing here runs longer than 5 ms
    others = [(max(s, a), min(e, b), n) for a, b, n in ev[lo:] if a < e and b > s and needle not in n]
    durs.append((e - s) / 1e3)
    n_other.append(len(others))
    if not others:
        alone += 1
    for a, b, n in others:
        shared_ns[n.split('(')[0][:90]] += b - a
durs.sort()
print(f'{len(mine)} launches of *{needle}*: duration median {durs[len(durs) // 2]:.1f} us (min {durs[0]:.1f}, max {durs[-1]:.1f}); '
      f'{alone} ran with nothing else on the chip; kernels overlapping one launch: median {sorted(n_other)[len(n_other) // 2]}')
for n, ns in sorted(shared_ns.items(), key=lambda kv: -kv[1])[:12]:
    print('   %8.1f us per launch next to  %s' % (ns / 1e3 / len(mine), n))
