#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter over its dispatches."""
import csv, sys, collections
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        k = r['Kernel_Name']
        if pat and pat not in k: continue
        acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-28s n=%-4d mean=%.4g' % (c, len(v), sum(v)/len(v)))
