#!/usr/bin/env python
"""Dev helper: GPU-side time per probe shape from a rocprofv3 kernel trace.  tools/gemm_probe.py launches every shape
3+reps times in a row with other (torch) kernels in between, so runs of consecutive crg GEMM/conv/split-K dispatches are
one shape each; prints the summed kernel time per call.  Usage: trace_groups.py <kernel_trace.csv> <calls_per_shape>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls = int(sys.argv[2])
ours = lambda n: ("gemm_glds_kernel" in n) or ("gemm_kernel" in n) or ("splitk_reduce" in n)
groups, cur = [], []
for r in rows:
    if ours(r["Kernel_Name"]):
        cur.append(r)
    elif cur:
        groups.append(cur)
        cur = []
if cur:
    groups.append(cur)
for i, g in enumerate(groups):
    t = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g) / 1e3
    main = [r for r in g if "splitk" not in r["Kernel_Name"]]
    blocks = int(main[0]["Grid_Size_X"]) // int(main[0]["Workgroup_Size_X"])
    print(f"shape {i:2d}: {t / calls:8.1f} us/call  ({len(g)} dispatches, grid {blocks} x {main[0]['Workgroup_Size_X']} thr, lds {main[0]['LDS_Block_Size']})")
