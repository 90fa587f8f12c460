#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats, PMC counter collection) into the small text/JSON files
kept under profiles/.  Usage:
    python tools/summarize_rocprof.py stats <rocprof_dir> <out.txt>
    python tools/summarize_rocprof.py pmc <workload> <fetch_dir> <write_dir> <out.json> <kernel substring> [compulsory bytes]
    python tools/summarize_rocprof.py sq <out.json> <kernel substring> <terms per launch> <pass_dir> [<pass_dir> ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit("no *{0} under {1}".format(suffix, d))
    return hits[-1]                                        # the newest run when a directory was reused


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
    }
    # gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE tallies a 128-byte request as 64 bytes, so streams that reach
    # HBM as 128-byte requests (vector loads) read x2 of the counter, while 64-byte requests (the s_load_dwordx16 record
    # reads of this kernel) read x1.  Which of the two a launch consists of is calibrated on a byte count that is known:
    # the kernel must read the whole record array exactly once (compulsory_bytes).
    factor = 2.0
    if compulsory:
        entry["compulsory_bytes"] = float(compulsory)
        entry["fetch_raw_over_compulsory"] = entry["fetch_bytes_raw"] / float(compulsory)
        factor = 1.0 if entry["fetch_raw_over_compulsory"] > 0.75 else 2.0
    entry["fetch_correction_factor"] = factor
    entry["fetch_bytes"] = (entry["fetch_bytes_raw"] or 0) * factor
    entry["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB units (x1024).  Correction factor "
                     "of FETCH_SIZE (see the comment in tools/summarize_rocprof.py): 2 when the records arrive through the "
                     "vector path (the software prefetch of mcd_math.h: RecordPrefetch touches every line first, 128-byte "
                     "requests; the scalar loads then hit in L2), 1 when they are fetched by the 64-byte scalar loads "
                     "themselves (option prefetch = 0, and every profile before the prefetch existed); chosen by which of "
                     "the two reproduces the compulsory read of the record array.")
    entry["traffic_bytes_per_launch"] = entry["fetch_bytes"] + (entry["write_bytes"] or 0)
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[workload] = entry["traffic_bytes_per_launch"]
    data[workload + "_detail"] = entry
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


def sq(out, kernel_sub, terms, dirs):
    """Mean SQ counter values per launch of the named kernel (tools/sq_counters.sh passes) + derived figures."""
    mean, dur = {}, []
    for d in dirs:
        acc = defaultdict(list)
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in acc.items():
            mean[k] = sum(v) / len(v)
    terms = float(terms)
    n_cu = 256
    cu_cycles = mean["SQ_BUSY_CU_CYCLES"]                 # busy cycles summed over the CUs
    us = sum(dur) / len(dur) / 1e3
    derived = {
        "kernel_us_under_counters": us,
        "shader_clock_GHz": cu_cycles / n_cu / (us * 1e3),
        # SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD (the gfx94x VALUBusy formula: x 4 / SIMDs per CU / busy cycles)
        "VALUBusy_percent": 100.0 * mean["SQ_ACTIVE_INST_VALU"] * 4 / 4 / cu_cycles,
        "valu_instructions_per_term": mean["SQ_INSTS_VALU"] * 64 / terms,
        "f64_fma_mul_add_trans_per_term": (mean["SQ_INSTS_VALU_FMA_F64"] + mean["SQ_INSTS_VALU_MUL_F64"] +
                                           mean["SQ_INSTS_VALU_ADD_F64"] + mean["SQ_INSTS_VALU_TRANS_F64"]) * 64 / terms,
        "int32_valu_per_term": mean["SQ_INSTS_VALU_INT32"] * 64 / terms,
        "lds_instructions_per_term": mean["SQ_INSTS_LDS"] * 64 / terms,
        "smem_instructions_per_wave": mean["SQ_INSTS_SMEM"] / mean["SQ_WAVES"],
        "lds_active_fraction_of_cu_cycles": mean["SQ_LDS_IDX_ACTIVE"] / cu_cycles,
        "lds_bank_conflict_fraction_of_lds_active": mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"] if mean["SQ_LDS_IDX_ACTIVE"] else 0.0,
    }
    data = {"kernel": kernel_sub, "terms_per_launch": terms, "counters_mean_per_launch": mean, "derived": derived,
            "note": "rocprofv3 --pmc, three passes of 8 SQ counters (tools/sq_counters.sh); values are sums over the chip"}
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:8])
