#!/usr/bin/env python3
"""Check rocprofv3's FETCH_SIZE / WRITE_SIZE against KNOWN byte counts in the access patterns of the scatter / gather kernels.

usage: pmc_calibration.py <op> <BxCxNxR> <fetch.json> <write.json>      (the JSONs: tools/pmc_by_kernel.py --json --by-grid over a
       `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python tools/opbench.py --ops <op> --shapes <shape>` run)
Prints, for the op's kernel, the bytes the launch MUST read / write (inputs far larger than L2, every element touched once), the
counters, and the factor that maps the counter onto the bytes.  MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide
(16 B / lane) coalesced streaming reads on gfx950 -- this is the check of that factor for 8-byte entry loads (segsum_tile) and the
two-row gather."""
import json
import sys

op, shape = sys.argv[1], [int(v) for v in sys.argv[2].split('x')]
fb, wb = json.load(open(sys.argv[3])), json.load(open(sys.argv[4]))
b, c, n, r = shape
s = r ** 3
need = {'devox_fwd': ('gather_lds', 4 * b * (c * s + 3 * n), 4 * b * c * n + 64 * b * n, 'grid + coordinates', 'outputs + inds / wgts'),
        'devox_bwd_apply': ('segsum_tile', 4 * b * c * n + 8 * b * 8 * n + 8 * b * s, 4 * b * c * s, 'grad rows + 8-byte plan entries + seg list', 'the whole grid'),
        'vox_apply': ('segsum_tile', 4 * b * c * n + 8 * b * n + 8 * b * s, 4 * b * c * s, 'feature rows + 8-byte plan entries + seg list', 'the whole grid'),
        'vox_bwd': ('gather_lds', 4 * b * (c * min(n, s) + n + min(n, s)), 4 * b * c * n, 'touched grid rows + ind + cnt', 'outputs')}[op]
needle, rd, wr, rwhat, wwhat = need
for key, vf in fb.items():
    if needle in key and key in wb:
        f_kib, w_kib = vf['FETCH_SIZE'], wb[key]['WRITE_SIZE']
        print(json.dumps({'op': op, 'BCNR': shape, 'kernel': key[:100], 'dispatches': vf['dispatches'],
                          'must_read_MB': round(rd / 1e6, 2), 'reads': rwhat, 'FETCH_SIZE_MB': round(f_kib * 1024 / 1e6, 2),
                          'read_bytes_per_FETCH_SIZE_byte': round(rd / (f_kib * 1024), 3),
                          'must_write_MB': round(wr / 1e6, 2), 'writes': wwhat, 'WRITE_SIZE_MB': round(w_kib * 1024 / 1e6, 2),
                          'written_bytes_per_WRITE_SIZE_byte': round(wr / (w_kib * 1024), 3)}))
