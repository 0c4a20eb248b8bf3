#!/usr/bin/env python3
"""Duration histogram per kernel from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
rows = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        rows[m.group(1) if m else r["Kernel_Name"][:30]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
edges = [10, 20, 50, 100, 200, 500, 1000, 1e9]
print("%-18s %6s %9s | time (ms) in launches of <10 <20 <50 <100 <200 <500 <1000 >=1000 us" % ("kernel", "calls", "total_ms"))
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    h = [0.0] * len(edges); c = [0] * len(edges)
    for x in v:
        for i, e in enumerate(edges):
            if x < e:
                h[i] += x / 1e3; c[i] += 1
                break
    print("%-18s %6d %9.2f | " % (k, len(v), sum(v) / 1e3) + " ".join("%5.1f(%d)" % (a, b) for a, b in zip(h, c)))
