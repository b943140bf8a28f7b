#!/usr/bin/env python3
"""Fill the @@NAME@@ placeholders of DESIGN.md / README.md from the evidence set of a round (gpurun_out/<round>/ or profiles/<round>_*).
usage: fill_design_numbers.py <dir> [prefix]      e.g.  fill_design_numbers.py profiles r04_"""
import json
import os
import re
import sys

d, pre = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
J = lambda name: json.load(open(os.path.join(d, pre + name)))
b205, b100 = J('bench_20_5.json'), J('bench.json')
steady = open(os.path.join(d, pre + 'bench_steady_state.txt')).readline()
ksum = re.search(r'kernel time ([0-9.]+) ms/step', steady).group(1)
r, m = b205['roofline'], b205['roofline_mfma']
vals = {
    'V205': f"{b205['value']:.1f}", 'MS205': f"{b205['ms_per_step']:.3f}", 'V100': f"{b100['value']:.1f}", 'MS100': f"{b100['ms_per_step']:.3f}",
    'KSUM': ksum, 'VETA': f"{J('bench_eager_torch_adam.json')['value']:.0f}", 'VE': f"{J('bench_eager.json')['value']:.0f}",
    'ROOF_LIVE': f"{r['avg_us']:.1f}", 'ROOF_GRAPH': f"{r['in_graph_us']}", 'ROOF_FRAC': f"{r['frac']:.3f}", 'ROOF_ADD': f"{r['with_fused_addend']['algorithmic_MB'] * 1e6 / (r['priced_on_us'] * 1e-6) / 8e12:.2f}",
    'TRAFFIC': f"{(r.get('traffic') or 0) / 1e6:.1f}",
    'MFMA_LIVE': f"{m['avg_us']:.1f}", 'MFMA_GRAPH': f"{m['in_graph_us']}", 'MFMA_FRAC': f"{m['frac']:.3f}", 'MFMA_EFF': f"{m['effective_fp32_TFLOPs']:.0f}",
    'CPU': f"{b205['cpu_baseline']['value']:.2f}",
}
for c in ('3', '4', '5'):
    bc = J(f'bench_cfg{c}.json')
    vals['V' + c], vals['MS' + c] = f"{bc['value']:.0f}", f"{bc['ms_per_step']:.2f}"
for path in ('DESIGN.md', 'README.md'):
    s = open(path).read()
    for k, v in vals.items():
        s = s.replace('@@' + k + '@@', v)
    left = set(re.findall(r'@@([A-Z0-9_]+)@@', s))
    if left:
        print(path, 'unfilled:', sorted(left))
    open(path, 'w').write(s)
print(vals)
