#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small files kept under profiles/.

    python tools/profile_summary.py <round-tag> <kernel-trace dir> <fetch dir> <write dir>

Writes profiles/<tag>_kernel_stats.csv (the --stats table), profiles/<tag>_pmc.json (HBM
traffic per launch of every kernel, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE is
in KiB-ish units of 1024 B and under-counts streaming reads by exactly 2x on gfx950 -- verified
here on a known 16.0 MB read with both 16-B and 8-B per-lane loads; WRITE_SIZE is exact) and
refreshes profiles/pmc_latest.json, which bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_means(d):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    out = collections.defaultdict(list)
    if not f:
        return {}
    for r in csv.DictReader(open(f[0])):
        out[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    tag, kt, fetch, write = sys.argv[1:5]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = glob.glob(os.path.join(kt, "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
    fetch_kb, write_kb = counter_means(fetch), counter_means(write)
    per_kernel = {}
    for k in sorted(set(fetch_kb) | set(write_kb)):
        rd = 2.0 * 1024.0 * fetch_kb.get(k, 0.0)        # gfx950: FETCH_SIZE counts half the bytes
        wr = 1024.0 * write_kb.get(k, 0.0)
        per_kernel[k] = {"fetch_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr,
                         "FETCH_SIZE_raw": fetch_kb.get(k), "WRITE_SIZE_raw": write_kb.get(k)}
    durations = {}
    for r in csv.DictReader(open(stats)):
        durations[r["Name"].split("(")[0].replace("void ", "").strip()] = {
            "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
    dom = [k for k in per_kernel if "k_gauss_cols" in k or "k_gauss_rows" in k]
    out = {"tag": tag, "units": "bytes per launch", "kernels": per_kernel, "durations": durations,
           "dominant_kernel": dom[0] if dom else None}
    for name in ("%s_pmc.json" % tag, "pmc_latest.json"):
        with open(os.path.join(ROOT, "profiles", name), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1)[:1500])


if __name__ == "__main__":
    main()
