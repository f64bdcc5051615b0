#!/usr/bin/env python
"""Dev helper for A/B runs: print the headline numbers of gpurun_out/ab_<tag>.json (a bench.py line)."""
import json, sys
for tag in sys.argv[1:]:
    d = json.load(open(f"gpurun_out/ab_{tag}.json"))
    print(tag, d["value"], d["ms_per_step"], d.get("kernel_families_ms_per_step"), flush=True)
