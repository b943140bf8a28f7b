#!/usr/bin/env python3
"""Markdown table of the matrix kernels' MFMA-pipe utilisation from two rocprofv3 --pmc passes (tools/pmc_by_kernel.py --json --by-grid):
   set 1: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
   set 2: SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
usage: pmc_mfma_table.py <set1.json> <set2.json> [min MFMAs per launch]

Columns.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GUI / 8 x 1024 pipes): GRBM_GUI_ACTIVE is summed over the 8 XCDs, the chip has
256 CUs x 4 SIMDs.  EFFECTIVE CLOCK = (GUI / 8) / launch duration of the same pass: what the chip clocked at under this kernel (2.4 GHz
nominal) -- the part of a low busy figure that is the power envelope rather than the schedule.  busy at 2.4 GHz = busy x clock / 2.4:
the utilisation against the chip's nominal peak (what `roofline_mfma.frac` prices).
SHORT LAUNCHES (< 60 us): GRBM_GUI_ACTIVE also counts the cycles the chip is active around a dispatch (launch ramp, tail), which is a
large share of a short launch -- (GUI / 8) / duration then reads 2.5-3.5 "GHz", above the 2.4 GHz the chip can clock.  Those rows are
marked `n/a (short)` in the clock and busy-at-2.4-GHz columns and must not be cited; their busy-of-cycles column
is still a ratio of two counters of the same pass and stands."""
SHORT_US = 60.0
import json
import sys

s1, s2 = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 5e4
print('| kernel @ launch grid | dispatches | MFMAs / launch | launch, shader cycles | launch, us | **effective clock, GHz** | **MFMA pipe busy** | '
      'busy at 2.4 GHz | LDS bank-conflict cycles / MFMA busy cycle | issue-stalled | issuing |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
rows = []
for k, a in s1.items():
    b = s2.get(k)
    if not b or a.get('SQ_INSTS_MFMA', 0) < floor or not b.get('GRBM_GUI_ACTIVE'):
        continue
    cyc = b['GRBM_GUI_ACTIVE'] / 8.0
    busy = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0)
    us = b.get('avg_us') or 0.0
    ghz = cyc / us / 1e3 if us else float('nan')
    wave = max(b.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    rows.append((a['SQ_VALU_MFMA_BUSY_CYCLES'] * a['dispatches'], k, a, b, cyc, us, ghz, busy, wave))
for _, k, a, b, cyc, us, ghz, busy, wave in sorted(rows, reverse=True):
    name = k.replace('void ', '').split('(')[0] + (' @' + k.split(' @grid=')[1] if ' @grid=' in k else '')
    short = us < SHORT_US or ghz > 2.45
    print(f"| `{name}` | {a['dispatches']} | {a['SQ_INSTS_MFMA'] / 1e6:.2f} M | {cyc / 1e3:.0f} k | {us:.1f} | "
          + ('n/a (short)' if short else f'**{ghz:.2f}**') + f" | **{busy:.2f}** | " + ('n/a (short)' if short else f'{busy * ghz / 2.4:.2f}') + " | "
          f"{a.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(a['SQ_VALU_MFMA_BUSY_CYCLES'], 1.0):.3f} | "
          f"{b.get('SQ_WAIT_INST_ANY', 0.0) / wave:.2f} | {b.get('SQ_ACTIVE_INST_ANY', 0.0) / wave:.2f} |")
