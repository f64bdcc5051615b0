#!/usr/bin/env python
"""Turn rocprofv3 outputs (gpurun_out/) into the committed summaries under profiles/.

  kernel stats : `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...`  -> *_kernel_stats.csv
  PMC passes   : `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` and `--pmc WRITE_SIZE ...` (separate passes, TCC slots)
                 -> *_counter_collection.csv

HBM traffic per launch follows MI355X_MICROARCH.md §HBM / cdna_hip_programming.md §7: the counters are in KiB and on
gfx950 FETCH_SIZE reports exactly half of a wide coalesced stream, so
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Usage: python tools/summarize_profile.py --round r01 --stats <kernel_stats.csv> [--fetch <counter.csv> --write <counter.csv>] [--bench-json <file>]
"""
import argparse
import collections
import csv
import json
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:90]


def pmc(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench-json")
    ap.add_argument("--cmd", default="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.stats)))
    dm = demangle([r["Name"] for r in rows])
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"# {a.round}: rocprofv3 --kernel-trace --stats summary", "", f"command: `rocprofv3 --kernel-trace --stats --output-format csv -- {a.cmd}` on 1x MI355X",
           f"total kernel time {total/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} dispatches", ""]
    if a.bench_json and os.path.exists(a.bench_json):
        out += ["bench.py line of the same run:", "```", open(a.bench_json).read().strip()[:3000], "```", ""]
    out += ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:28]:
        out.append(f"| `{short(dm[r['Name']])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    traffic = {}
    if a.fetch and a.write:
        f, w = pmc(a.fetch), pmc(a.write)
        dm2 = demangle(list(f.keys()))
        out += ["", "## HBM traffic per launch (PMC, separate passes)", "",
                "`hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024` (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md §HBM)", "",
                "| kernel | launches | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM MB/launch (corrected) |", "|---|---:|---:|---:|---:|"]
        names = sorted(f, key=lambda k: -(2 * f[k][0] + w.get(k, [0, 1])[0]))
        for k in names[:16]:
            fl = f[k][0] / max(1, f[k][1])
            wl = w[k][0] / max(1, w[k][1]) if k in w else 0.0
            mb = (2 * fl + wl) * 1024 / 1e6
            traffic[short(dm2[k])] = dict(launches=f[k][1], fetch_kib_raw=fl, write_kib=wl, hbm_bytes_per_launch=(2 * fl + wl) * 1024)
            out.append(f"| `{short(dm2[k])}` | {f[k][1]} | {fl:.1f} | {wl:.1f} | {mb:.2f} |")
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    open(os.path.join(REPO, "profiles", f"{a.round}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
    if traffic:
        json.dump(traffic, open(os.path.join(REPO, "profiles", f"{a.round}_traffic.json"), "w"), indent=1)
    print("\n".join(out[-22:]))


if __name__ == "__main__":
    main()
