#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name (reads *counter_collection.csv under a directory)."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if pat and pat not in k:
            continue
        a = acc[k[:90]][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f'    {c:32s} avg {s / n:16.1f}   dispatches {n}')
