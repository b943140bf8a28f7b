#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name (reads *counter_collection.csv under a directory).
usage: pmc_by_kernel.py <dir> [name-filter] [--json]   (--json: one JSON object {kernel: {counter: avg, 'dispatches': n}})"""
import collections
import csv
import glob
import json
import sys

args = [a for a in sys.argv[1:] if a != '--json']
as_json = '--json' in sys.argv
root, pat = args[0], (args[1] if len(args) > 1 else '')
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if pat and pat not in k:
            continue
        a = acc[k[:160]][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
if as_json:
    print(json.dumps({k: dict({c: s / n for c, (s, n) in cs.items()}, dispatches=max(n for _, n in cs.values())) for k, cs in acc.items()}))
else:
    for k, cs in acc.items():
        print(k[:90])
        for c, (s, n) in sorted(cs.items()):
            print(f'    {c:32s} avg {s / n:16.1f}   dispatches {n}')
