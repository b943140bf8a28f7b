#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel (reads *counter_collection.csv under a directory).

usage: pmc_by_kernel.py <dir> [name-filter] [--json] [--by-grid]
  --by-grid : one entry per (kernel name, launch grid) instead of per kernel name -- the launches of one template at different layer
              shapes are not averaged together (key "<name> @grid=<threads>")
  --json    : one JSON object {key: {counter: avg, 'dispatches': n, 'avg_us': mean launch duration in THIS (profiled) run}}
The duration comes from the Start / End timestamps of the same rows: with GRBM_GUI_ACTIVE in the pass, GUI / 8 / avg_us is the
effective shader clock of the kernel (the counter is summed over the 8 XCDs)."""
import collections
import csv
import glob
import json
import sys

flags = {a for a in sys.argv[1:] if a.startswith('--')}
args = [a for a in sys.argv[1:] if not a.startswith('--')]
root, pat = args[0], (args[1] if len(args) > 1 else '')
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if pat and pat not in k:
            continue
        key = k[:160]
        if '--by-grid' in flags:
            key += ' @grid=' + str(row.get('Grid_Size', '?'))
        a = acc[key][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
        try:
            d = dur[(key, row['Counter_Name'])]
            d[0] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3; d[1] += 1
        except (KeyError, ValueError):
            pass


def avg_us(key, counters):
    c = next(iter(counters))
    s, n = dur.get((key, c), (0.0, 0))
    return round(s / n, 3) if n else None


if '--json' in flags:
    print(json.dumps({k: dict({c: s / n for c, (s, n) in cs.items()}, dispatches=max(n for _, n in cs.values()), avg_us=avg_us(k, cs))
                      for k, cs in acc.items()}))
else:
    for k, cs in acc.items():
        print(k[:110], f'  [avg {avg_us(k, cs)} us]')
        for c, (s, n) in sorted(cs.items()):
            print(f'    {c:32s} avg {s / n:16.1f}   dispatches {n}')
