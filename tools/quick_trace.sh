#!/bin/bash
# Kernel-trace summary of one bench configuration (no counter passes): per-kernel calls, total and
# average time. usage: tools/quick_trace.sh <outdir> <bench args...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py "$@" > "$OUT/trace.json" 2> "$OUT/trace.err" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
d = sys.argv[1]
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(d, "kernel_stats.txt"), "w") as out:
        for r in rows[:16]:
            m = re.search(r"(k_\w+)(<[^>]*>)?", r["Name"])
            name = (m.group(1) + (m.group(2) or "")) if m else r["Name"][:40]
            out.write("%-30s %6d %10.2f ms %9.1f us %6.2f %%\n" % (name, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf "$OUT/trace"
cat "$OUT/kernel_stats.txt"
