#!/usr/bin/env python
"""Per-kernel means of the counters of one rocprofv3 --pmc pass (counter_collection.csv): python tools/pmc_summary.py <csv> [name filter]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:80]
    if flt and flt not in k:
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = len(disp[k])
    print('%s   (%d dispatches)' % (k, n))
    for c, x in sorted(v.items()):
        print('   %-28s %.4g per dispatch' % (c, x / n))
    if 'TCC_HIT_sum' in v and 'TCC_MISS_sum' in v:
        h, m = v['TCC_HIT_sum'], v['TCC_MISS_sum']
        print('   L2 hit rate %.3f; requests %.2f GB, misses %.2f GB per dispatch (128-byte lines)' % (h / (h + m), (h + m) / n * 128 / 1e9, m / n * 128 / 1e9))
    if 'SQ_WAVE_CYCLES' in v:
        wc = v['SQ_WAVE_CYCLES']
        print('   of the wave cycles: ' + ', '.join('%s %.0f %%' % (c[3:], 100 * v[c] / wc) for c in ('SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY') if c in v))
