#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC averages collect_evidence.sh wrote (FETCH_SIZE / WRITE_SIZE in KiB, separate
passes).  bench.py reads it for roofline.traffic: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 correction)."""
import json
import os
import sys

d = sys.argv[1]


def load(name):
    try:
        return json.load(open(os.path.join(d, name)))
    except (OSError, ValueError):
        return {}


def pick(table, *needles):
    for k, v in table.items():
        if all(n in k for n in needles):
            return k, v
    return None, None


rows = []
fb, wb = load('pmc_FETCH_SIZE_bench.json'), load('pmc_WRITE_SIZE_bench.json')
# the fused devoxelize gather of the R=32 stage inside the training step: 1024-thread, BatchNorm+LeakyReLU transform
# (the software-pipelined kernel since the end of round 2; the classic one in older traces or with PVCNN_GATHER_PIPE=0)
kf, vf = pick(fb, 'gather_lds_pipe_kernel', 'TrilinearFromCoords', 'XfBnAct')
kw, vw = pick(wb, 'gather_lds_pipe_kernel', 'TrilinearFromCoords', 'XfBnAct')
if not (vf and vw):
    kf, vf = pick(fb, 'gather_lds_kernel', 'TrilinearFromCoords', '1024', 'XfBnAct')
    kw, vw = pick(wb, 'gather_lds_kernel', 'TrilinearFromCoords', '1024', 'XfBnAct')
if vf and vw:
    rows.append({'op': 'trilinear_devoxelize_fwd', 'shape_BCNR': [16, 64, 4096, 32], 'kernel_name': kf[:120], 'where': 'inside bench.py steps',
                 'FETCH_SIZE_KiB': round(vf['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vw['WRITE_SIZE'], 1), 'dispatches': vf['dispatches']})
for shp in ('16x64x4096x16', '16x128x4096x16', '16x64x4096x32'):
    fo, wo = load(f'pmc_FETCH_SIZE_opbench_{shp}.json'), load(f'pmc_WRITE_SIZE_opbench_{shp}.json')
    for op, needles in (('trilinear_devoxelize_bwd', ('segsum_tile_kernel',)), ('trilinear_devoxelize_fwd (op-level, unfused)', ('gather_lds_', 'TrilinearFromCoords'))):
        kf, vf = pick(fo, *needles)
        kw, vw = pick(wo, *needles)
        if vf and vw:
            rows.append({'op': op, 'shape_BCNR': [int(v) for v in shp.split('x')], 'kernel_name': kf[:120], 'where': 'tools/opbench.py',
                         'FETCH_SIZE_KiB': round(vf['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vw['WRITE_SIZE'], 1), 'dispatches': vf['dispatches']})
print(json.dumps({'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace (separate passes), tools/collect_evidence.sh',
                  'units': 'KiB per launch, averaged over dispatches; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950',
                  'kernels': rows}, indent=1))
