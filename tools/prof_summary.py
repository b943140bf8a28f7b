#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (name, calls, total ms, avg us, %)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot / 1e6:.3f} ms over {len(rows)} distinct kernels')
for r in rows[:top]:
    print('%9.3f ms %6.2f%% calls=%5s avg=%9.1f us  %s' % (int(r['TotalDurationNs']) / 1e6, 100.0 * int(r['TotalDurationNs']) / tot,
                                                         r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:150]))
