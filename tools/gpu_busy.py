#!/usr/bin/env python
"""Dev helper: GPU busy fraction over the tail of a rocprofv3 kernel trace (sum of kernel durations / wall span) and the
largest idle gaps.  Usage: gpu_busy.py <kernel_trace.csv> <n_last_dispatches>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  = {100*busy/span:.1f} %")
pos = [g for g in gaps if g[0] > 0]
print(f"idle total {sum(g[0] for g in pos)/1e6:.2f} ms in {len(pos)} gaps; gaps > 20 us: {sum(1 for g in pos if g[0] > 20000)} totalling {sum(g[0] for g in pos if g[0] > 20000)/1e6:.2f} ms")
for g in sorted(pos, reverse=True)[:8]:
    print(f"  {g[0]/1e3:8.1f} us  after {g[1]}  before {g[2]}")
