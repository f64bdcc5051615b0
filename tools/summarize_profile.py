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


def slot_of(name):
    """rocprofv3 kernel name (mangled, or mis-demangled: its demangler prints `__bf16, true` as `bool _Accum, bool, E`)
    -> cremage_amd._lib.SLOT_NAMES entry (enum crg_kernel_slot), or None for kernels that are not ours."""
    import re
    # gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, NS, PAIR>: parse the arguments by position (PAIR is a bool too)
    m = re.search(r"gemm_glds_kernelILi(\d)E(?:DF16b|DF16_|f)Lb([01])ELi\d+ELi\d+ELi\d+ELi(\d+)ELb[01]E", name)
    if m:
        wnt, conv, ns = m.group(1), m.group(2) == "1", m.group(3)
    else:
        m = re.search(r"gemm_glds_kernel<(\d), (bool _Accum, bool, E|[\w ]+, (?:true|false)), \d+, \d+, \d+, (\d+), (?:true|false)(?:, (?:true|false|\d))?>", name)
        if m:
            wnt, ns = m.group(1), m.group(3)
            conv = m.group(2).startswith("bool _Accum") or m.group(2).endswith("true")
    if m:
        if ns == "2":  # the split-plane fp32-class variant is accounted in the x3 slots
            return "conv_x3" if conv else "gemm_x3"
        return ("conv_w" if conv else "gemm_w") + wnt
    # conv3_rowhalo_kernel<WNT, YT, PAIR, NS, KG, MT>
    m = re.search(r"conv3_rowhalo_kernelILi(\d)E(?:DF16b|DF16_|f)Lb[01]ELi(\d+)E", name) or \
        re.search(r"conv3_rowhalo_kernel<(\d), (?:bool _Accum, bool, E|[\w ]+, (?:true|false)), (\d+),", name)
    if m:
        return "conv_x3" if m.group(2) == "2" else "conv_w" + m.group(1)
    m = re.search(r"conv3_ring_kernelILi(\d)E", name) or re.search(r"conv3_ring_kernel<(\d),", name)
    if m:
        return "conv_w" + m.group(1)
    m = re.search(r"conv3_pp_kernelILi(\d)E", name) or re.search(r"conv3_pp_kernel<(\d),", name)
    if m:
        return "conv_w" + m.group(1)
    m = re.search(r"gemm_ring_kernelILi(\d)E", name) or re.search(r"gemm_ring_kernel<(\d),", name)
    if m:
        return "gemm_w" + m.group(1)
    if "lngemm_kernel" in name:
        return "lngemm"
    if "gemm_kernel" in name:
        return "conv_x3" if (("Lb1E" in name) or re.search(r", true[,>]", name)) else "gemm_x3"
    for pat, slot in (("splitk_reduce", "splitk_reduce"), ("attn_kernel", "attention"), ("attn_sp_kernel", "attention"), ("attn_dma_kernel", "attention"), ("attn_ctx_kernel", "attention"),
                      ("gn_stats", "gn_stats"), ("gn_finalize", "gn_stats"), ("gn_apply", "gn_apply"), ("gn_small", "gn_apply"),
                      ("layernorm_kernel", "layernorm"), ("softmax_rows", "softmax"), ("conv_small", "conv_small")):
        if pat in name:
            return slot
    return None


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
    # per kernel slot (the unit bench.py's `roofline` reports): calls / avg duration from the stats pass, HBM bytes from the PMC passes
    by_slot = collections.defaultdict(lambda: dict(calls=0, total_ms=0.0, fetch_kib=0.0, write_kib=0.0, pmc_launches=0))
    for r in rows:
        sl = slot_of(r["Name"])
        if sl:
            by_slot[sl]["calls"] += int(r["Calls"])
            by_slot[sl]["total_ms"] += float(r["TotalDurationNs"]) / 1e6
    if a.fetch and a.write:
        for k, (v, n) in f.items():
            sl = slot_of(k)
            if sl:
                by_slot[sl]["fetch_kib"] += v
                by_slot[sl]["pmc_launches"] += n
        for k, (v, n) in w.items():
            sl = slot_of(k)
            if sl:
                by_slot[sl]["write_kib"] += v
    out += ["", "## Per kernel slot (enum crg_kernel_slot; what bench.py's `roofline` reports)", "",
            "| slot | calls | total ms | avg us | HBM MB/launch (PMC, corrected) |", "|---|---:|---:|---:|---:|"]
    slots_json = {}
    for sl, e in sorted(by_slot.items(), key=lambda kv: -kv[1]["total_ms"]):
        hb = (2 * e["fetch_kib"] + e["write_kib"]) * 1024 / e["pmc_launches"] if e["pmc_launches"] else None
        slots_json[sl] = dict(calls=e["calls"], total_ms=round(e["total_ms"], 3), avg_us=round(1e3 * e["total_ms"] / max(1, e["calls"]), 2),
                              hbm_bytes_per_launch=hb)
        out.append(f"| {sl} | {e['calls']} | {e['total_ms']:.2f} | {1e3 * e['total_ms'] / max(1, e['calls']):.1f} | {'' if hb is None else f'{hb / 1e6:.2f}'} |")
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    json.dump(slots_json, open(os.path.join(REPO, "profiles", f"{a.round}_by_slot.json"), "w"), indent=1)
    open(os.path.join(REPO, "profiles", f"{a.round}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
    if traffic:
        json.dump(traffic, open(os.path.join(REPO, "profiles", f"{a.round}_traffic.json"), "w"), indent=1)
    print("\n".join(out[-40:]))


if __name__ == "__main__":
    main()
