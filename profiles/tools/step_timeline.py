"""Kernel timeline of one steady-state reverse step from a rocprofv3 kernel trace:
    python profiles/tools/step_timeline.py <kernel_trace.csv> [step index]
Prints start offset, idle gap before the kernel, duration and name, then the step span and the summed gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_sample_update' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = prev = int(rows[a]['End_Timestamp'])
gaps = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gaps += max(s - prev, 0)
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
    prev = e
print(f'step span {(prev - t0) / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us')
