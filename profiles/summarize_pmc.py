"""Summarise rocprofv3 --pmc CSVs (per-dispatch counter rows) into per-kernel averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        a = agg[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
    if 'kpd::k_egnn' not in k and 'kpd::k_node' not in k and 'kpd::k_gvp' not in k and 'kpd::k_proj' not in k:
        continue
    print(k)
    for c, (tot, n) in sorted(agg[k].items()):
        print(f'   {c:28s} avg/dispatch {tot / n:16.1f}   dispatches {n}')
    m = agg[k]
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        busy = m['SQ_VALU_MFMA_BUSY_CYCLES'][0] / m['SQ_VALU_MFMA_BUSY_CYCLES'][1]
        gui = m['GRBM_GUI_ACTIVE'][0] / m['GRBM_GUI_ACTIVE'][1]
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
        print(f'   MfmaUtil ~ {100 * busy / (gui / 8 * 1024):.1f} %  (MFMA busy cycles / (active cycles x 1024 SIMDs))')
    if 'FETCH_SIZE' in m:
        kb = m['FETCH_SIZE'][0] / m['FETCH_SIZE'][1]
        print(f'   FETCH_SIZE avg {kb:.0f} KB/dispatch -> x2 gfx950 correction = {2 * kb / 1024:.1f} MB')
    if 'WRITE_SIZE' in m:
        kb = m['WRITE_SIZE'][0] / m['WRITE_SIZE'][1]
        print(f'   WRITE_SIZE avg {kb:.0f} KB/dispatch = {kb / 1024:.1f} MB')
