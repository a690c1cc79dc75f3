#!/usr/bin/env python
"""Print the headline and the kernel table of a bench.py JSON line, largest time share first."""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('%.2f img/s  %.2f ms/step  whole-step MFMA frac %.4f  hash %s' % (j['value'], j['ms_per_step'], j['mfma_peak_frac_whole_step'], j.get('source_hash', '')[:12]))
for k, v in sorted(j.get('kernels', {}).items(), key=lambda kv: -kv[1]['ms_per_step']):
    print('%-28s n=%3d  %7.3f ms  %8.1f us  %s TFLOP/s' % (k, v['launches_per_step'], v['ms_per_step'], v['avg_us'], ('%6.1f' % v['tflops']) if v.get('tflops') else '     -'))
if j.get('roofline'):
    print('roofline', json.dumps(j['roofline']))
