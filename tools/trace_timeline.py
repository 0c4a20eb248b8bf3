#!/usr/bin/env python3
"""Timeline of the last frame in a rocprofv3 --kernel-trace CSV directory: per launch its
duration and the idle gap before it. usage: trace_timeline.py <dir> [frame]   (frame: -1 the last, -2 the one before ...;
bench.py's last frame is its short untimed counters frame, the timed ones come before it)"""
import csv, glob, os, re, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24]))
rows.sort()
# a frame ends with its k_frame_resolve
ends = [i for i, r in enumerate(rows) if r[2] == "k_frame_resolve"]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
bounds = [0] + [e + 1 for e in ends]
frames = [(bounds[k], bounds[k + 1]) for k in range(len(bounds) - 1)]
i0, i1 = frames[which] if frames else (0, len(rows))
sel = rows[i0:i1]
t0 = sel[0][0]
busy = sum(e - s for s, e, _ in sel)
print("frame: %d launches, span %.2f ms, kernels %.2f ms, gaps %.2f ms" % (
    len(sel), (sel[-1][1] - t0) / 1e6, busy / 1e6, (sel[-1][1] - t0 - busy) / 1e6))
prev = t0
for s, e, n in sel:
    print("%9.1f us  +%6.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
