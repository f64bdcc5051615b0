#!/usr/bin/env python
"""Dev helper: average a PMC counter over runs of consecutive dispatches of our GEMM/conv kernels
(tools/gemm_probe.py launches every shape 3+reps times in a row).  Usage: pmc_groups.py <counter_collection.csv> <group_len>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_glds_kernel" in r["Kernel_Name"] or "gemm_kernel" in r["Kernel_Name"]]
g = int(sys.argv[2])
names = sorted({r["Counter_Name"] for r in rows})
for cn in names:
    rr = [r for r in rows if r["Counter_Name"] == cn]
    rr.sort(key=lambda r: int(r["Dispatch_Id"]))
    for i in range(0, len(rr), g):
        grp = rr[i:i + g]
        v = sum(float(r["Counter_Value"]) for r in grp) / len(grp)
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / len(grp) / 1e3
        print(f"{cn:16s} group {i // g}: grid {grp[0]['Grid_Size']:>8s}  avg {v:14.1f}   ({len(grp)} launches, {dur:.1f} us under PMC)")
