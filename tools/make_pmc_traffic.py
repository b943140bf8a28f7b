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
# every other scatter / gather kernel of the step: averages over the launches of one kernel template (several shapes per step),
# for the traffic / algorithmic ratio of the family -- the per-shape algorithmic bytes are in the bench line's `kernels`
for op, needles in (('trilinear_devoxelize_fwd at R = 16 (3 launches per step: C = 64, 64, 128)', ('gather_lds_pipe_rows_kernel', 'TrilinearFromCoords')),
                    ('trilinear_devoxelize_bwd + avg_voxelize_fwd applies (8 launches per step)', ('segsum_tile_kernel',)),
                    ('avg_voxelize_bwd (3 launches per step)', ('gather_lds_kernel', 'VoxelMean'))):
    kf, vf = pick(fb, *needles)
    kw, vw = pick(wb, *needles)
    if vf and vw:
        rows.append({'op': op, 'kernel_name': kf[:120], 'where': 'inside bench.py steps (average over the launches of this template)',
                     'FETCH_SIZE_KiB': round(vf['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vw['WRITE_SIZE'], 1), 'dispatches': vf['dispatches']})
print(json.dumps({'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace (separate passes), tools/collect_evidence.sh',
                  'units': 'KiB per launch, averaged over dispatches; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950',
                  'kernels': rows}, indent=1))
