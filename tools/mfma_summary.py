#!/usr/bin/env python3
"""Counter-based MFMA utilisation per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE):

    python tools/mfma_summary.py --build r02a --out profiles/mfma.json DIR

utilisation = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 * 1024): the MfmaUtil formula of rocprofiler-sdk's
counter_defs.yaml (busy cycles summed over the chip's 1024 SIMDs over the cycles the GPU was active; rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back").  On dispatches shorter than ~0.3 ms
GRBM_GUI_ACTIVE reads high (same section), i.e. the utilisation reads LOW; `util_wall` therefore also divides the busy
cycles by the dispatch's wall time at the 2.4 GHz peak clock (a lower bound on the busy fraction of the real clock).
SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = MFMA flops issued (counter_defs.yaml)."""
import argparse
import collections
import csv
import glob
import json
import re

from pmc_summary import ENTRY

N_SIMD, N_XCD, PEAK_HZ = 1024, 8, 2.4e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--build', default='?')
    ap.add_argument('dirs', nargs='+')
    a = ap.parse_args()
    rows = collections.defaultdict(lambda: collections.defaultdict(dict))     # entry -> dispatch id -> counter -> value
    dur = collections.defaultdict(dict)
    kname = {}
    for d in a.dirs:
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                for pat, entry in ENTRY:
                    if re.search(pat, row['Kernel_Name']):
                        did = (f, row['Dispatch_Id'])
                        rows[entry][did][row['Counter_Name']] = float(row['Counter_Value'])
                        dur[entry][did] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
                        kname[entry] = row['Kernel_Name'][:100]
                        break
    out = {'build': a.build, 'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)', 'per_kernel': {}}
    for entry, disp in sorted(rows.items()):
        busy = [c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for c in disp.values()]
        if not busy or max(busy) == 0:
            continue
        gui = [c.get('GRBM_GUI_ACTIVE', 0.0) for c in disp.values()]
        mops = [c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) for c in disp.values()]
        sqb = [c.get('SQ_BUSY_CYCLES', 0.0) for c in disp.values()]
        ns = [dur[entry][k] for k in disp]
        n = len(busy)
        rec = {'launches': n, 'kernel': kname[entry], 'avg_ns_under_pmc': sum(ns) / n,
               'SQ_VALU_MFMA_BUSY_CYCLES': sum(busy) / n, 'GRBM_GUI_ACTIVE': sum(gui) / n, 'SQ_BUSY_CYCLES': sum(sqb) / n,
               'mfma_flops_issued': sum(mops) / n * 512,
               'util': sum(busy) / (sum(gui) / N_XCD * N_SIMD) if sum(gui) else None,
               'util_wall': sum(busy) / (sum(ns) * 1e-9 * PEAK_HZ * N_SIMD)}
        out['per_kernel'][entry] = rec
        print('%-28s %4d launches  %8.1f us  MFMA busy %5.1f %% of GUI-active cycles, %5.1f %% of wall @2.4 GHz, %.2f GFLOP issued'
              % (entry, n, rec['avg_ns_under_pmc'] / 1e3, 100 * (rec['util'] or 0), 100 * rec['util_wall'], rec['mfma_flops_issued'] / 1e9))
    if a.out:
        json.dump(out, open(a.out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
