#!/usr/bin/env python
"""Dev helper: per-kernel sums of every counter in a rocprofv3 counter_collection.csv (kernels whose name contains argv[2])."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:60], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key, r["Counter_Name"]] += 1
for key, d in agg.items():
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {v / cnt[key, c]:16.1f} per launch ({cnt[key, c]} launches)")
