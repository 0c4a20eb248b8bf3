#!/usr/bin/env python3
"""Timeline of the last frame in a rocprofv3 --kernel-trace CSV directory: per launch its
duration and the idle gap before it. usage: trace_timeline.py <dir> [n_last_launches]"""
import csv, glob, os, re, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24]))
rows.sort()
# the last frame starts at the last k_set_counts that follows a k_frame_resolve (or the first one)
starts = [i for i, r in enumerate(rows) if r[2] == "k_set_counts" and (i == 0 or any(q[2] == "k_frame_resolve" for q in rows[max(0, i - 12):i]))]
i0 = starts[-1] if starts else 0
sel = rows[i0:]
t0 = sel[0][0]
busy = sum(e - s for s, e, _ in sel)
print("last frame: %d launches, span %.2f ms, kernels %.2f ms, gaps %.2f ms" % (
    len(sel), (sel[-1][1] - t0) / 1e6, busy / 1e6, (sel[-1][1] - t0 - busy) / 1e6))
prev = t0
for s, e, n in sel:
    print("%9.1f us  +%6.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
