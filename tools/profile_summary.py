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


def all_counters(dirs):
    """kernel -> counter -> mean value per launch, over several --pmc pass directories."""
    out = collections.defaultdict(dict)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                acc[(r["Kernel_Name"].split("(")[0].replace("void ", "").strip(), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in acc.items():
                out[k][c] = sum(v) / len(v)
    return out


def main():
    if sys.argv[1] == "sq":
        # python tools/profile_summary.py sq <tag> <dir with sq1 .. sqN> : profiles/<tag>_sq_counters.json
        tag, base = sys.argv[2:4]
        counters = all_counters(sorted(glob.glob(os.path.join(base, "sq*"))))
        doc = {"how": "rocprofv3 --pmc <set> -- python3 bench.py (tools/collect_profiles.sh), one pass per set of <= 4 "
                      "counters; mean per launch, summed over the device as rocprofv3 reports them; SQ_*_CYCLES / "
                      "SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles", "kernels": counters}
        for k, c in counters.items():
            if "SQC_DCACHE_REQ" in c and c["SQC_DCACHE_REQ"] > 0:
                c["scalar_cache_hit_rate"] = c.get("SQC_DCACHE_HITS", 0.0) / c["SQC_DCACHE_REQ"]
            if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
                c["wave_time_fraction_in_s_waitcnt"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                c["wave_time_fraction_waiting_for_issue"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        with open(os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag), "w") as f:
            json.dump(doc, f, indent=1)
        print(json.dumps({k: v for k, v in counters.items() if "gauss" in k or "muse" in k}, indent=1)[:3000])
        return
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
    dom = sorted((k for k in per_kernel if "k_gauss_cols" in k or "k_gauss_rows" in k or "k_muse_rows" in k),
                 key=lambda k: -durations.get(k, {}).get("calls", 0) * durations.get(k, {}).get("avg_ns", 0))
    out = {"tag": tag, "units": "bytes per launch", "kernels": per_kernel, "durations": durations,
           "dominant_kernel": dom[0] if dom else None}
    for name in ("%s_pmc.json" % tag, "pmc_latest.json"):
        with open(os.path.join(ROOT, "profiles", name), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1)[:1500])


if __name__ == "__main__":
    main()
