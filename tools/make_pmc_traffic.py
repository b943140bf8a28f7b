#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-(kernel, launch grid) PMC averages collect_evidence.sh wrote (FETCH_SIZE / WRITE_SIZE in KiB,
separate passes, tools/pmc_by_kernel.py --by-grid).  bench.py reads it for roofline.traffic: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes
per launch (gfx950 correction of FETCH_SIZE for wide streaming reads; the factor is checked against known byte counts by the
`calibration` entries: the same counters on the op-level benchmark, where the algorithmic bytes of a launch are known)."""
import json
import os
import sys

d = sys.argv[1]
# argv[2] (optional): a bench.py JSON line of the same code -- the headline launch's shape is read from its roofline entry instead of being
# assumed
headline_shape = None
if len(sys.argv) > 2:
    try:
        headline_shape = json.load(open(sys.argv[2]))['roofline']['shape_BCNR']
    except (OSError, ValueError, KeyError, TypeError):
        headline_shape = None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from pvcnn_amd._lib import sources_digest
    digest = sources_digest()
except Exception:                                        # noqa: BLE001
    digest = None


def load(name):
    try:
        return json.load(open(os.path.join(d, name)))
    except (OSError, ValueError):
        return {}


fb, wb = load('pmc_FETCH_SIZE_bench.json'), load('pmc_WRITE_SIZE_bench.json')
FAMILIES = (('trilinear_devoxelize_fwd', ('gather_lds_pipe_kernel', 'TrilinearFromCoords')),
            ('trilinear_devoxelize_fwd (two-row kernel, R = 16)', ('gather_lds_pipe_rows_kernel', 'TrilinearFromCoords')),
            ('trilinear_devoxelize_fwd (generic kernel)', ('gather_lds_kernel', 'TrilinearFromCoords')),
            ('avg_voxelize_bwd', ('gather_lds_kernel', 'VoxelMean')),
            ('avg_voxelize_fwd / trilinear_devoxelize_bwd applies', ('segsum_tile_kernel',)),
            ('scatter plans', ('csr_',)))
rows = []
for key, vf in sorted(fb.items()):
    vw = wb.get(key)
    if not vw:
        continue
    op = next((name for name, needles in FAMILIES if all(n in key for n in needles)), None)
    if op is None:
        continue
    name, _, grid = key.partition(' @grid=')
    row = {'op': op, 'kernel_name': name[:120], 'grid_threads': int(grid) if grid.isdigit() else None, 'where': 'inside bench.py steps',
           'FETCH_SIZE_KiB': round(vf['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vw['WRITE_SIZE'], 1), 'dispatches': vf['dispatches']}
    if op == 'trilinear_devoxelize_fwd' and 'XfBnAct' in name and headline_shape:      # the headline launch: the largest-R stage of the step
        row['shape_BCNR'] = headline_shape
    rows.append(row)
calib = []
fo, wo = load('pmc_FETCH_SIZE_opbench.json'), load('pmc_WRITE_SIZE_opbench.json')
for key, vf in sorted(fo.items()):
    vw = wo.get(key)
    if vw and any(n in key for n in ('gather_lds', 'segsum_tile')):
        name, _, grid = key.partition(' @grid=')
        calib.append({'kernel_name': name[:120], 'grid_threads': int(grid) if grid.isdigit() else None,
                      'FETCH_SIZE_KiB': round(vf['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vw['WRITE_SIZE'], 1), 'dispatches': vf['dispatches']})
print(json.dumps({'sources_digest': digest, 'trace_commit': os.environ.get('PVCNN_TRACE_COMMIT'), 'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace (separate passes) -- python bench.py ..., tools/collect_evidence.sh',
                  'units': 'KiB per launch, averaged over dispatches; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950',
                  'kernels': rows,
                  'calibration': {'what': 'the same counters over tools/opbench.py (one op at one shape per launch: algorithmic bytes known, inputs '
                                          'far larger than the caches), see profiles/README.md for the read-off', 'launches': calib}}, indent=1))
