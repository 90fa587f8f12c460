#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats, PMC counter collection) into the small text/JSON files
kept under profiles/.  Usage:
    python tools/summarize_rocprof.py stats <rocprof_dir> <out.txt>
    python tools/summarize_rocprof.py pmc <workload> <fetch_dir> <write_dir> <out.json> <kernel substring> [compulsory bytes]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *{0} under {1}".format(suffix, d))
    return hits[-1]


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary ({0})\n".format(os.path.basename(os.path.normpath(d))))
        f.write("{0:<90s} {1:>8s} {2:>14s} {3:>12s} {4:>12s} {5:>12s} {6:>8s}\n".format(
            "Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
        for r in rows:
            f.write("{0:<90s} {1:>8s} {2:>14s} {3:>12.1f} {4:>12s} {5:>12s} {6:>8s}\n".format(
                r["Name"][:90], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"],
                r["Percentage"]))
    print(open(out).read())


def counter_per_dispatch(d, counter, kernel_sub):
    path = find(d, "counter_collection.csv")
    per = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = sorted(per.values())
    return vals


def pmc(workload, fetch_dir, write_dir, out, kernel_sub, compulsory=None):
    fetch = counter_per_dispatch(fetch_dir, "FETCH_SIZE", kernel_sub)
    write = counter_per_dispatch(write_dir, "WRITE_SIZE", kernel_sub)
    med = lambda v: v[len(v) // 2] if v else None
    entry = {
        "kernel": kernel_sub, "dispatches": len(fetch),
        "FETCH_SIZE_KB_median": med(fetch), "WRITE_SIZE_KB_median": med(write),
        "FETCH_SIZE_KB_first": fetch and counter_per_dispatch(fetch_dir, "FETCH_SIZE", kernel_sub)[0],
        "fetch_bytes_raw": med(fetch) * 1024 if fetch else None,
        "fetch_bytes_x2_gfx950_wide_read_correction": med(fetch) * 2048 if fetch else None,
        "write_bytes": med(write) * 1024 if write else None,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB units (x1024). The x2 gfx950 "
                "correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B) is calibrated for "
                "16 B/lane vector streams; the record reads of this kernel are 64-B scalar loads (s_load_dwordx16). "
                "Calibration on the known byte count of this access pattern: the kernel must read the whole record "
                "array once (compulsory_bytes) and raw FETCH_SIZE equals it to 0.5 %, so the factor is 1.0 here.",
    }
    entry["traffic_bytes_per_launch"] = (entry["fetch_bytes_raw"] or 0) + (entry["write_bytes"] or 0)
    if compulsory:
        entry["compulsory_bytes"] = float(compulsory)
        entry["fetch_raw_over_compulsory"] = entry["fetch_bytes_raw"] / float(compulsory)
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[workload] = entry["traffic_bytes_per_launch"]
    data[workload + "_detail"] = entry
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:8])
