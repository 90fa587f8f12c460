"""Scratch: kernel durations and the idle time before each kernel from a rocprofv3 --kernel-trace CSV (one stream):
    python tools/chain_trace_summary.py gpurun_out/chain_trace/*/*_kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = []
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void |mcd::", "", r["Kernel_Name"]).split("(")[0][:70]))
rows.sort()
dur, gap = defaultdict(list), defaultdict(list)
for i, (s, e, name) in enumerate(rows):
    dur[name].append(e - s)
    if i:
        gap[name].append(s - rows[i - 1][1])
print("{0:72s} {1:>7s} {2:>10s} {3:>12s}".format("kernel", "calls", "avg us", "idle before"))
for name in sorted(dur, key=lambda k: -sum(dur[k])):
    g = sorted(gap[name])
    print("{0:72s} {1:7d} {2:10.2f} {3:12.2f}".format(name, len(dur[name]), sum(dur[name]) / len(dur[name]) / 1e3,
                                                      (g[len(g) // 2] if g else 0) / 1e3))
