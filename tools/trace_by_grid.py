#!/usr/bin/env python
"""Dev helper: average GPU-side duration per (kernel, grid size) from a rocprofv3 kernel trace CSV - separates the shapes
a probe script runs through one kernel.  Usage: trace_by_grid.py <kernel_trace.csv> [name-substring ...]"""
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
        continue
    short = n.split("(")[0].replace("void (anonymous namespace)::", "")[:60]
    k = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    a = agg.setdefault(k, [0, 0])
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[1] += 1
for (n, gx, gy), (t, c) in agg.items():
    print(f"{n:60s} grid {gx:6d} x {gy:3d}: {c:5d} calls, avg {t / c / 1e3:8.2f} us")
