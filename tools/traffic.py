#!/usr/bin/env python
"""HBM traffic per launch of each kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the read side is doubled before the two are added.  Writes profiles/<out>.json:
{class: {"fetch_kib": .., "write_kib": .., "hbm_bytes_per_launch": .., "launches": ..}}.
"""
import csv, json, sys, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import denseASPP  # noqa: F401
from semantic_segmentation_amd import profiling

def klass(name):
    n = name.replace(' ', '')
    for c in profiling.CLASSES[1:]:
        stem = c[:-1] + ',' if c.endswith('>') else c
        if stem in n or c + '(' in n or (not c.endswith('>') and c in n):
            return c
    return None

def collect(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter: continue
            k = klass(r['Kernel_Name'])
            if k: acc[k].append(float(r['Counter_Value']))
    return acc

fetch = collect(sys.argv[1], 'FETCH_SIZE'); write = collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch[k]) / max(len(fetch[k]), 1); w = sum(write[k]) / max(len(write[k]), 1)
    out[k] = dict(fetch_kib=round(f, 1), write_kib=round(w, 1), hbm_bytes_per_launch=int((2 * f + w) * 1024), launches=len(fetch[k]))
if len(sys.argv) > 4:      # per-GPU batch of the profiled bench.py run: bench.py reports the traffic only for that batch
    from semantic_segmentation_amd import _lib
    out['_meta'] = dict(batch=int(sys.argv[4]), source_hash=_lib.source_hash())
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
